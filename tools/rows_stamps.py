"""Phase stamps of the row-list data gradient of the first Conv3D (k_igemm<1,...> over the voxel rows), R200k sweep."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from lisec_amd import Constants, _lib, ops
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer

dev = torch.device("cuda")
lib = _lib.load()
lib.lisec_debug_igemm_stamps.argtypes = [ctypes.c_void_p]
net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints)
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                Constants.ny // 2, Constants.nz)
ycls = torch.zeros(100, 200, 2, device=dev)
yreg = torch.zeros(100, 200, 14, device=dev)
c = net.layers[0]["conv"]
for name in sys.argv[1:] or ["r200k"]:
    cloud = bench.u20k_cloud(0) if name == "u20k" else bench.r200k_cloud(0)
    sample = vox(torch.from_numpy(cloud).to(dev))
    net.train_step(sample, ycls, yreg)
    dg = net.dgeom[c.name]
    rows = (sample.coords, sample.info, max(sample.cap, 1))
    for _ in range(3):
        ops.conv_forward(dg, net.dact["mid1.z"], net.packed_t[c.name][0], net.dout_rows, rows=rows, queue=net.rows_queue)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_igemm_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    ops.conv_forward(dg, net.dact["mid1.z"], net.packed_t[c.name][0], net.dout_rows, rows=rows, queue=net.rows_queue)
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[(t[:, 0] > 0) & (t[:, 4] > 0)]
    t0 = t[:, 0].min()
    print(f"{name}: {len(t)} workgroups ran to the end; starts 0 .. {(t[:, 0].max() - t0) / 100:.1f} us, last end "
          f"{(t[:, 4].max() - t0) / 100:.1f} us")
    steps = t[:, 5]
    for lo, hi in ((0, 0), (1, 6), (7, 10), (11, 14), (15, 20), (21, 27)):
        q = t[(steps >= lo) & (steps <= hi)]
        if len(q) == 0:
            continue
        d = lambda a, b: np.median((q[:, b] - q[:, a]) / 100.0)
        loop = (q[:, 3] - q[:, 2]) / 100.0
        print(f"   {len(q):5d} workgroups with {lo:2d}-{hi:2d} steps: setup {d(0, 1):5.2f}  first tile {d(1, 2):5.2f}  main loop "
              f"{np.median(loop):6.2f} ({np.median(loop / np.maximum(q[:, 5], 1)):.2f} us/step)  epilogue {d(3, 4):5.2f}  whole {d(0, 4):6.2f}")
