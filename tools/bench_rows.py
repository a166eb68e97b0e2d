"""The row-list contractions of the first Conv3D's backward alone (data gradient at the voxels, weight gradient over the
voxel rows), U20k and R200k sweeps (GPU box only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lisec_amd import Constants, ops
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer
from tools.bench_field import timed

if __name__ == "__main__":
    dev = torch.device("cuda")
    net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints)
    vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                    Constants.ny // 2, Constants.nz)
    ycls = torch.zeros(100, 200, 2, device=dev)
    yreg = torch.zeros(100, 200, 14, device=dev)
    c = net.layers[0]["conv"]
    for name, cloud in (("u20k", bench.u20k_cloud(0)), ("r200k", bench.r200k_cloud(0))):
        sample = vox(torch.from_numpy(cloud).to(dev))
        net.train_step(sample, ycls, yreg)
        torch.cuda.synchronize()
        dg = net.dgeom[c.name]
        rows = (sample.coords, sample.info, max(sample.cap, 1))
        dz = net.dact["mid1.z"]
        delta = net.vfe.saved_field("delta")
        dW = torch.empty(27, 64, 64, device=dev)
        t_w = timed(lambda: ops.conv_wgrad(dg, dz, delta, dW, net.wgrad_ws, transpose_out=True, rows=rows), 20)
        t_d = timed(lambda: ops.conv_forward(dg, dz, net.packed_t[c.name][0], net.dout_rows, rows=rows, queue=net.rows_queue), 20)
        V = sample.host_info()["V"]
        gf = 2.0 * V * 13.5 * 64 * 64 / 1e9
        print(f"{name}: V {V} ({gf:.2f} GFLOP over the (voxel, tap) pairs that exist): row-list weight gradient {t_w:.1f} us, "
              f"row-list data gradient {t_d:.1f} us", flush=True)
