"""Micro-benchmark of the implicit-GEMM kernel on the Lyft-grid layer shapes (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lisec_amd import ops

DEV = "cuda"
PEAK = 157.3e12


def run(name, mode, ind, outd, k, s, p, cin, cout, iters=10, in_bn=False):
    x = torch.randn(*ind, cin, device=DEV)
    ntaps = k[0] * k[1] * k[2]
    w = torch.randn(ntaps, cin, cout, device=DEV) * 0.05
    wp = ops.pack_weights(w, ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*outd, cout, device=DEV)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    st = torch.zeros(ops.num_mblocks(g), 2, cout, dtype=torch.float64, device=DEV)
    bn = torch.randn(4 * cin, device=DEV) if in_bn else None
    for _ in range(2):
        ops.conv_forward(g, x, wp, out, in_bn=bn, flags=ops.IN_RELU if in_bn else 0, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_forward(g, x, wp, out, in_bn=bn, flags=ops.IN_RELU if in_bn else 0, stats=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = outd[0] * outd[1] * outd[2]
    fl = 2.0 * M * ntaps * cin * cout
    print(f"{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s  {100*fl/ms/1e-3/PEAK:5.1f}% of fp32-MFMA peak", flush=True)


def run_wgrad(name, mode, ind, outd, k, s, p, cin, cout, iters=10, in_bn=False):
    x = torch.randn(*ind, cin, device=DEV)
    dy = torch.randn(*outd, cout, device=DEV)
    ntaps = k[0] * k[1] * k[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=DEV)
    dW = torch.empty(ntaps, cin, cout, device=DEV)
    bn = torch.randn(4 * cin, device=DEV) if in_bn else None
    fl_ = ops.IN_RELU if in_bn else 0
    for _ in range(2):
        ops.conv_wgrad(g, x, dy, dW, ws, in_bn=bn, flags=fl_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_wgrad(g, x, dy, dW, ws, in_bn=bn, flags=fl_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = outd[0] * outd[1] * outd[2]
    fl = 2.0 * M * ntaps * cin * cout
    print(f"{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s  {100*fl/ms/1e-3/PEAK:5.1f}% of fp32-MFMA peak", flush=True)


if __name__ == "__main__":
    run_wgrad("mid1 wgrad (dense form)", 0, (8, 200, 400), (4, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    run_wgrad("mid2 wgrad", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64)
    run_wgrad("mid3 wgrad", 0, (2, 200, 400), (1, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    run_wgrad("rpn1.conv1 wgrad", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=True)
    run_wgrad("rpn2.conv1 wgrad", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=True)
    run_wgrad("rpn3.conv1 wgrad", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, in_bn=True)
    run("mid1 conv3d s(2,1,1)", 0, (8, 200, 400), (4, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    run("mid2 conv3d s(1,1,1)", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64)
    run("mid3 conv3d s(2,1,1)", 0, (2, 200, 400), (1, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    run("mid1 dense64", 0, (4, 200, 400), (4, 200, 400), (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64, in_bn=True)
    run("mid1 dgrad", 1, (4, 200, 400), (8, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    run("rpn1.conv0 s2 64->128", 0, (1, 200, 400), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128)
    run("rpn1.conv1 128->128", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=True)
    run("rpn2.conv1 128->128", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, in_bn=True)
    run("rpn3.conv1 256->256", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, in_bn=True)
    run("up1 deconv k3s1", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256, in_bn=True)
    run("up2 deconv k2s2", 1, (1, 50, 100), (1, 100, 200), (1, 2, 2), (1, 2, 2), (0, 0, 0), 128, 256, in_bn=True)
    run("up3 deconv k4s4", 1, (1, 25, 50), (1, 100, 200), (1, 4, 4), (1, 4, 4), (0, 0, 0), 256, 256, in_bn=True)
    run("heads 768->16", 0, (1, 100, 200), (1, 100, 200), (1, 1, 1), (1, 1, 1), (0, 0, 0), 768, 16)
