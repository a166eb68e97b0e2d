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
elif which == "vfe":
    import numpy as np
    from lisec_amd.params import ParamStore
    from lisec_amd.vfe import VFEStack
    from lisec_amd.voxelizer import Voxelizer
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-55, 55, 20000), rng.uniform(-55, 55, 20000), rng.uniform(-0.5, 2.5, 20000)], 1).astype(np.float32)
    dev = torch.device("cuda")
    vfe = VFEStack(ParamStore(dev))
    sample = Voxelizer(0.5, 0.25, 0.25, 35, 100, 200, 8)(pts)
    grid = torch.empty(8, 200, 400, 64, device=dev)
    for _ in range(it):
        vfe.forward(sample, True, out=grid)
        vfe.rewrite_grid(grid)
    torch.cuda.synchronize()
