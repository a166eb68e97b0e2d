"""Phase timing inside the weight-gradient kernels (100 MHz stamps of every workgroup) on the Lyft layer shapes.
usage: python tools/wgrad_stamps.py [name filter ...]   (LISEC_TUNING=wgrad_blocks=512 to try another split target)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from lisec_amd import _lib, ops
from bench_wgrad_plans import CASES

dev = "cuda"
lib = _lib.load()


def case(name, mode, ind, outd, k, s, p, cin, cout, in_bn, iters=10):
    x = torch.randn(*ind, cin, device=dev)
    dy = torch.randn(*outd, cout, device=dev)
    ntaps = k[0] * k[1] * k[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=dev)
    dW = torch.empty(ntaps, cin, cout, device=dev)
    bn = torch.randn(4 * cin, device=dev) if in_bn else None
    run = lambda: ops.conv_wgrad(g, x, dy, dW, ws, in_bn=bn, flags=ops.IN_RELU if in_bn else 0, transpose_out=mode == 1)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    M = outd[0] * outd[1] * outd[2]
    flops = 2.0 * M * ntaps * cin * cout
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_wgrad_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_wgrad_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    print(f"{name}: {us:6.1f} us per call (kernel + slab reduce) = {flops / us / 1e6 / 157.3:.2f} of peak; {len(t)} workgroups", flush=True)
    if not len(t):
        return
    if os.environ.get("STAMPS_DUMP"):
        np.save(os.path.join(os.environ["STAMPS_DUMP"], "wgrad_" + name.replace(" ", "_") + ".npy"), t)
    t0 = t[:, 0].min()
    T = lambda k: (t[:, k] - t0) / 100.0
    cu = ((t[:, 6] >> 32) & 0xf) * 256 + ((t[:, 6] >> 8) & 0xff)
    ids, counts = np.unique(cu, return_counts=True)
    print(f"   {len(ids)} CUs; workgroups per CU over the launch: " + ", ".join(f"{k}: {int((counts == k).sum())}" for k in sorted(set(counts))))
    print(f"   starts: p50 {np.median(T(0)):.1f} p90 {np.percentile(T(0), 90):.1f} max {T(0).max():.1f} us; ends: p10 {np.percentile(T(3), 10):.1f} "
          f"p50 {np.median(T(3)):.1f} p90 {np.percentile(T(3), 90):.1f} max {T(3).max():.1f} us")
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 3], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0])
    print(f"   mean workgroups alive {np.sum(alive[:-1] * dur) / max(dur.sum(), 1):.0f}, peak {int(alive.max())}")
    nt = np.maximum(t[:, 5], 1)
    d = lambda a, b: (t[:, b] - t[:, a]) / 100.0
    print(f"   per workgroup (median): entry -> first tile in LDS {np.median(d(0, 1)):.2f} us; loop {np.median(d(1, 2)):.2f} us over "
          f"{np.median(t[:, 5]):.0f} tiles = {np.median(d(1, 2) / nt):.2f} us per tile; slab stores {np.median(d(2, 3)):.2f} us; whole {np.median(d(0, 3)):.2f}")
    if (t[:, 7] > t[:, 4]).all():
        ghz = (t[:, 7] - t[:, 4]) / np.maximum(t[:, 2] - t[:, 1], 1) / 10.0
        print(f"   in-kernel clock over the loop (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz, p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}")


if __name__ == "__main__":
    only = sys.argv[1:]
    print("tuning:", _lib.get_tuning())
    for c in CASES:
        if not only or any(o in c[0] for o in only):
            case(*c)
