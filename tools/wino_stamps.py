"""Phase timing inside k_wino (100 MHz stamps + shader-cycle counts of every workgroup) at the Lyft geometries."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lisec_amd import _lib, ops

dev = "cuda"
lib = _lib.load()


def case(name, mode, ind, outd, KD, sd, pd, cin, cout, flags=0):
    x = torch.randn(*ind, cin, device=dev)
    w = torch.randn(KD * 9, cin, cout, device=dev) * 0.05
    wu = ops.pack_weights_winograd(w, KD, cin, cout, cin * cout, cout, 1, flip=(mode == 1))
    out = torch.empty(*outd, cout, device=dev)
    g = ops.geom(mode, ind, outd, (KD, 3, 3), (sd, 1, 1), (pd, 1, 1), cin, cout)
    for _ in range(20):
        ops.conv_forward_winograd(g, x, wu, out, flags=flags)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_wino_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    ops.conv_forward_winograd(g, x, wu, out, flags=flags)
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_wino_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"{name} (flags {flags:#x}): {len(t)} workgroups; starts 0 .. {(t[:, 0].max() - t0) / 100:.1f} us, last end {(t[:, 3].max() - t0) / 100:.1f} us")
    for nch in sorted(set(t[:, 6])):
        q = t[t[:, 6] == nch]
        d = lambda a, b: np.median((q[:, b] - q[:, a]) / 100.0)
        loop = (q[:, 2] - q[:, 1]) / 100.0
        cyc = (q[:, 5] - q[:, 4]).astype(np.float64)
        ghz = np.median(cyc / np.maximum(loop, 1e-9)) / 1e3
        print(f"   {len(q):5d} workgroups with {int(nch):3d} chunks: prologue {d(0, 1):5.2f}  K loop {np.median(loop):6.2f} us "
              f"({np.median(loop) / max(nch, 1):.3f} us = {np.median(cyc) / max(nch, 1):.0f} cycles per chunk, clock {ghz:.2f} GHz)  "
              f"epilogue {d(2, 3):5.2f}  whole {d(0, 3):6.2f} us")
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 3], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0])
    print(f"   mean workgroups alive {np.sum(alive[:-1] * dur) / max(dur.sum(), 1):.0f}")


if __name__ == "__main__":
    H, W = 200, 400
    for fl in ((0,) if len(sys.argv) > 1 else (0, 0x10000, 0x40000, 0x50000, 0x60000)):
        case("mid2 fwd", 0, (4, H, W), (2, H, W), 3, 1, 0, 64, 64, fl)
    case("mid3 fwd", 0, (2, H, W), (1, H, W), 3, 2, 1, 64, 64)
    case("mid3 dgrad", 1, (1, H, W), (2, H, W), 3, 2, 1, 64, 64)
    case("mid2 dgrad", 1, (2, H, W), (4, H, W), 3, 1, 0, 64, 64)
    case("rpn1.conv1 fwd", 0, (1, 100, 200), (1, 100, 200), 1, 1, 0, 128, 128)
