"""Weight-gradient launches alone on the Lyft layer shapes, for several split targets (lisec_tuning.wgrad_blocks)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lisec_amd import _lib, ops

DEV = "cuda"
PEAK = 157.3e12
CASES = [
    ("mid2 wgrad", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, False),
    ("mid3 wgrad", 0, (2, 200, 400), (1, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, False),
    ("mid1.dense wgrad", 0, (4, 200, 400), (4, 200, 400), (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64, True),
    ("rpn1.conv0 wgrad", 0, (1, 200, 400), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128, False),
    ("rpn1.conv1 wgrad", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True),
    ("rpn2.conv1 wgrad", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True),
    ("rpn3.conv1 wgrad", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, True),
    ("up1 wgrad", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256, True),
]


def run(name, mode, ind, outd, k, s, p, cin, cout, in_bn, iters=10):
    x = torch.randn(*ind, cin, device=DEV)
    dy = torch.randn(*outd, cout, device=DEV)
    ntaps = k[0] * k[1] * k[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=DEV)
    dW = torch.empty(ntaps, cin, cout, device=DEV)
    bn = torch.randn(4 * cin, device=DEV) if in_bn else None
    fl_ = ops.IN_RELU if in_bn else 0
    for _ in range(2):
        ops.conv_wgrad(g, x, dy, dW, ws, in_bn=bn, flags=fl_, transpose_out=mode == 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_wgrad(g, x, dy, dW, ws, in_bn=bn, flags=fl_, transpose_out=mode == 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = outd[0] * outd[1] * outd[2]
    fl = 2.0 * M * ntaps * cin * cout
    return ms * 1e3, fl / ms / 1e9 / 157.3e3


if __name__ == "__main__":
    targets = [int(a) for a in sys.argv[1:]] or [256, 384, 512, 768, 1024, 1536]
    print("%-22s" % "wgrad_blocks" + "".join("%16d" % t for t in targets))
    for c in CASES:
        row = "%-22s" % c[0]
        for t in targets:
            _lib.set_tuning(wgrad_blocks=t)
            us, frac = run(*c)
            row += "  %7.1f us %.2f" % (us, frac)
        print(row, flush=True)
