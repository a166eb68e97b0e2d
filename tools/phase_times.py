"""In-step phase durations WITHOUT a profiler: timing events recorded on the main stream inside the recorded step plan
(LisecNet._mark) and read back after every replay.  usage: python tools/phase_times.py [steps]   (LISEC_TUNING=... applies)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from lisec_amd import Constants, _lib
from lisec_amd.network import LisecNet, PipelinedStep
from lisec_amd.voxelizer import Voxelizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = _lib.require_gpu()
net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints, device=dev)
net.phase_marks = {}
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2, Constants.ny // 2,
                Constants.nz, device=dev)
cloud = bench.u20k_cloud(0) if os.environ.get("CLOUD", "u20k") == "u20k" else bench.r200k_cloud(0)
pts = torch.from_numpy(cloud).to(dev)
yc, yr = bench.synthetic_targets(0, net.Ho, net.Wo)
yc, yr = torch.from_numpy(yc).to(dev), torch.from_numpy(yr).to(dev)
step = PipelinedStep(net, vox, len(cloud), dtype=pts.dtype)
step.prime(pts, yc, yr)
step.stage_next(pts, yc, yr)
names = None
acc = []
nosync = bool(os.environ.get("NOSYNC"))     # steady state: the host runs ahead, only the LAST step's marks are read (one sample)
for k in range(steps + 3):
    step.step()
    if nosync and k < steps + 2:
        continue
    torch.cuda.synchronize()
    if names is None:
        names = list(net.phase_marks.keys())
    if k >= 3 or nosync:
        m = net.phase_marks
        acc.append([m[names[0]].elapsed_ms(m[n]) * 1e3 for n in names])
a = np.median(np.array(acc), 0)
order = np.argsort(a)
prev = 0.0
print(f"median over {steps} steps, us since '{names[0]}' (main stream; a mark completes when everything enqueued before it on "
      "the main stream has)")
for i in order:
    print(f"{a[i]:8.1f}  (+{a[i] - prev:7.1f})  {names[i]}")
    prev = a[i]
