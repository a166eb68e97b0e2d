"""Forward-only (inference BatchNormalization) latency of voxelise + VFE + middle + RPN on one sweep (GPU box only)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import u20k_cloud
from lisec_amd import Constants
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer

if __name__ == "__main__":
    dev = torch.device("cuda")
    net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints, device=dev)
    vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                    Constants.ny // 2, Constants.nz, device=dev)
    pts = torch.from_numpy(u20k_cloud(0)).to(dev)
    for _ in range(5):
        net.forward(vox(pts), training=False)
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        net.forward(vox(pts), training=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"voxelise + forward (inference BN): {1e3 * dt:.2f} ms per sweep = {1 / dt:.0f} sweeps/s")
