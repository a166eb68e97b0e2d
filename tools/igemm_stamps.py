"""Phase timing inside k_igemm_halo on the mid2 layer: forward vs data gradient (100 MHz stamps of every workgroup)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lisec_amd import _lib, ops

dev = "cuda"
lib = _lib.load()
lib.lisec_debug_igemm_stamps.argtypes = [ctypes.c_void_p]


def case(name, mode, ind, outd, k, s, p, cin, cout, extra=0):
    x = torch.randn(*ind, cin, device=dev)
    ntaps = k[0] * k[1] * k[2]
    w = torch.randn(ntaps, cin, cout, device=dev) * 0.05
    wp = ops.pack_weights(w, ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*outd, cout, device=dev)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    for _ in range(3):
        ops.conv_forward(g, x, wp, out, flags=ops.TAG_ROOFLINE | extra)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_igemm_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    ops.conv_forward(g, x, wp, out, flags=ops.TAG_ROOFLINE | extra)
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"{name}: {len(t)} workgroups; starts 0 .. {(t[:, 0].max() - t0) / 100:.1f} us, last end {(t[:, 4].max() - t0) / 100:.1f} us")
    for nsteps in sorted(set(t[:, 5])):
        q = t[t[:, 5] == nsteps]
        d = lambda a, b: np.median((q[:, b] - q[:, a]) / 100.0)
        loop = (q[:, 3] - q[:, 2]) / 100.0
        print(f"   {len(q):5d} workgroups with {int(nsteps):3d} steps: setup {d(0, 1):5.2f}  first tile in LDS {d(1, 2):5.2f}  "
              f"main loop {np.median(loop):6.2f} ({np.median(loop) / max(nsteps, 1):.2f} us/step)  epilogue {d(3, 4):5.2f}  "
              f"whole {d(0, 4):6.2f} us")
    # concurrency: workgroups alive over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 4], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0])
    print(f"   mean workgroups alive {np.sum(alive[:-1] * dur) / max(dur.sum(), 1):.0f}")
    if os.environ.get("STAMPS_TIMELINE"):
        span = int((t[:, 4].max() - t0) / 100) + 1
        for u in range(0, span, 20):
            lo, hi = t0 + u * 100, t0 + (u + 20) * 100
            started = int(((t[:, 0] >= lo) & (t[:, 0] < hi)).sum())
            ended = int(((t[:, 4] >= lo) & (t[:, 4] < hi)).sum())
            mid = lo + 1000
            live_now = int(((t[:, 0] <= mid) & (t[:, 4] > mid)).sum())
            print(f"      t {u:4d}-{u + 20:4d} us: started {started:4d} ended {ended:4d} alive at midpoint {live_now:4d}")


if __name__ == "__main__":
    case("mid2 fwd", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64)
    case("mid2 dgrad", 1, (2, 200, 400), (4, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64)
    case("mid1 fwd (dense form)", 0, (8, 200, 400), (4, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
