"""Phase timing inside the contraction kernels on the small RPN maps (100 MHz stamps of every workgroup), launched the
way the training forward launches them: BatchNormalization(+ReLU) on load, batch statistics into a sink, default plan."""
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


def case(name, mode, ind, outd, k, s, p, cin, cout, in_bn=True, sink=True, iters=20):
    x = torch.randn(*ind, cin, device=dev)
    ntaps = k[0] * k[1] * k[2]
    w = torch.randn(ntaps, cin, cout, device=dev) * 0.05
    wp = ops.pack_weights(w, ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*outd, cout, device=dev)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    M = outd[0] * outd[1] * outd[2]
    bn = torch.randn(4 * cin, device=dev) if in_bn else None
    sk = None
    if sink:
        gam, bet = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        sk = ops.BnSink(cout, M, dev, gamma=gam, beta=bet, bnstate=torch.zeros(4 * cout, device=dev))
    fl = ops.IN_RELU if in_bn else 0
    run = lambda: ops.conv_forward(g, x, wp, out, in_bn=bn, flags=fl, sink=sk)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    flops = 2.0 * M * ntaps * cin * cout
    # the same launch with COLD weights: a step runs every layer once, so no layer finds its kernel in L2; here 64 copies
    # of the packed kernel are rotated through (64 x a few MB: beyond every XCD's 4 MB)
    copies = [wp.clone() for _ in range(64)]
    for c in copies[:3]:
        ops.conv_forward(g, x, c, out, in_bn=bn, flags=fl, sink=sk)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        ops.conv_forward(g, x, copies[i % 64], out, in_bn=bn, flags=fl, sink=sk)
    e1.record()
    torch.cuda.synchronize()
    us_cold = e0.elapsed_time(e1) / iters * 1e3
    del copies
    print(f"   ({us_cold:.1f} us per call with a different copy of the packed kernel every call: weights not in L2)")
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_igemm_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    if os.environ.get("STAMPS_DUMP"):
        np.save(os.path.join(os.environ["STAMPS_DUMP"], "stamps_" + name.split(" (")[0].replace(" ", "_").replace(">", "") + ".npy"), t)
    print(f"{name}: {us:6.1f} us per call back to back = {flops / us / 1e6:5.1f} TF/s = {flops / us / 1e6 / 157.3:.2f} of peak; "
          f"{len(t)} stamped workgroups", flush=True)
    if not len(t):
        return
    t0 = t[:, 0].min()
    ended = t[t[:, 4] > 0]
    print(f"   starts 0 .. {(t[:, 0].max() - t0) / 100:.1f} us, last end {(ended[:, 4].max() - t0) / 100:.1f} us")
    ends = np.sort((ended[:, 4] - t0) / 100.0)
    print("   end-time percentiles (us): " + "  ".join(f"p{q}={np.percentile(ends, q):.1f}" for q in (10, 50, 90, 99, 100)))
    cu = ((t[:, 6] >> 32) & 0xf) * 256 + ((t[:, 6] >> 8) & 0xff)              # (xcc, se/sh/cu) of every workgroup
    ids, counts = np.unique(cu, return_counts=True)
    print(f"   {len(ids)} CUs used; workgroups per CU: " + ", ".join(f"{k}: {int((counts == k).sum())} CUs" for k in sorted(set(counts))))
    per_cu = dict(zip(ids, counts))
    occ = np.array([per_cu[c] for c in cu])
    for k in sorted(set(counts)):
        sel = (occ == k) & (t[:, 4] > 0)
        if sel.any():
            print(f"      workgroups on CUs holding {k}: median whole {np.median((t[sel, 4] - t[sel, 0]) / 100.0):6.2f} us, "
                  f"median end {np.median((t[sel, 4] - t0) / 100.0):6.2f}, last end {((t[sel, 4] - t0) / 100.0).max():6.2f}")
    for nsteps in sorted(set(ended[:, 5])):
        q = ended[ended[:, 5] == nsteps]
        d = lambda a, b: np.median((q[:, b] - q[:, a]) / 100.0)
        loop = (q[:, 3] - q[:, 2]) / 100.0
        print(f"   {len(q):5d} workgroups with {int(nsteps):3d} steps: setup {d(0, 1):5.2f}  first tile in LDS {d(1, 2):5.2f}  "
              f"main loop {np.median(loop):6.2f} ({np.median(loop) / max(nsteps, 1):.2f} us/step)  epilogue {d(3, 4):5.2f}  "
              f"whole {d(0, 4):6.2f} us")
        sl = q[q[:, 7] > 0]                                   # K-sliced tiles: slab store + ticket, then the last arriver's sum
        if len(sl):
            arrive = (sl[:, 7] - sl[:, 3]) / 100.0
            after = (sl[:, 4] - sl[:, 7]) / 100.0
            last = after > 0.5
            print(f"         K slices: slab store + ticket {np.median(arrive):5.2f} us; {int(last.sum())} last arrivers: sum of the slabs + "
                  f"epilogue {np.median(after[last]) if last.any() else 0:5.2f} us (max {after.max():5.2f})")


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        _case = case
        case = lambda name, *a, **k: _case(name, *a, **k) if any(o in name for o in only) else None
    print("tuning:", _lib.get_tuning())
    case("rpn1.conv1 128->128 (20000)", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128)
    case("rpn2.conv0 s2 128->128 (5000)", 0, (1, 100, 200), (1, 50, 100), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128)
    case("rpn2.conv1 128->128 (5000)", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128)
    case("rpn3.conv0 s2 128->256 (1250)", 0, (1, 50, 100), (1, 25, 50), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 256)
    case("rpn3.conv1 256->256 (1250)", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256)
    case("rpn3.conv1 dgrad (1250)", 1, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, in_bn=False, sink=False)
    case("rpn2.conv1 dgrad (5000)", 1, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=False, sink=False)
    case("rpn1.conv0 s2 dgrad (80000)", 1, (1, 100, 200), (1, 200, 400), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 64, in_bn=False, sink=False)
    case("rpn2.conv0 s2 dgrad (20000)", 1, (1, 50, 100), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128, in_bn=False, sink=False)
    case("rpn3.conv0 s2 dgrad (5000)", 1, (1, 25, 50), (1, 50, 100), (1, 3, 3), (1, 2, 2), (0, 1, 1), 256, 128, in_bn=False, sink=False)
    case("up1 deconv k3s1 128->256", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256, sink=False)
