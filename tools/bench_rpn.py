"""The RPN-shaped contractions alone (no statistics table, so the K slices may stay inside the workgroups).
usage: [LISEC_HALF_N=0] [LISEC_MAX_SPLITK=n] python tools/bench_rpn.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lisec_amd import ops

DEV = "cuda"


def run(name, mode, ind, outd, k, s, p, cin, cout, iters=30, in_bn=True):
    x = torch.randn(*ind, cin, device=DEV)
    ntaps = k[0] * k[1] * k[2]
    w = torch.randn(ntaps, cin, cout, device=DEV) * 0.05
    wp = ops.pack_weights(w, ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*outd, cout, device=DEV)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    bn = torch.randn(4 * cin, device=DEV) if in_bn else None
    fl_ = ops.IN_RELU if in_bn else 0
    for _ in range(3):
        ops.conv_forward(g, x, wp, out, in_bn=bn, flags=fl_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_forward(g, x, wp, out, in_bn=bn, flags=fl_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = outd[0] * outd[1] * outd[2]
    fl = 2.0 * M * ntaps * cin * cout
    print(f"{name:30s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    run("rpn1.conv0 s2 64->128", 0, (1, 200, 400), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128, in_bn=False)
    run("rpn1.conv1 128->128", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128)
    run("rpn1 K x2 (256->128)", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 128)
    run("rpn2.conv1 128->128", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128)
    run("rpn2 K x2 (256->128)", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 128)
    run("rpn3.conv1 256->256", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256)
    run("rpn3 K x2 (512->256)", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 512, 256)
    run("rpn1 dgrad 128->128", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=False)
    run("rpn2.conv0 dgrad s2", 1, (1, 50, 100), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128, in_bn=False)
    run("up1 deconv k3s1 128->256", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256)
