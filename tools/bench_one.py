"""Run one Lyft-shape layer kernel a few times (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lisec_amd import ops
from tools.bench_conv import run  # noqa
which = sys.argv[1] if len(sys.argv) > 1 else "mid1"
it = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if which == "mid1":
    run("mid1 conv3d s(2,1,1)", 0, (8, 200, 400), (4, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=it)
elif which == "mid1_dgrad":
    run("mid1 dgrad", 1, (4, 200, 400), (8, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=it)
elif which == "rpn3":
    run("rpn3.conv1 256->256", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, iters=it, in_bn=True)
