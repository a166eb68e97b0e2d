"""How long does the HOST take to enqueue one training step (vs the GPU time of the step)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import u20k_cloud, synthetic_targets
from lisec_amd import Constants
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer

dev = torch.device("cuda")
net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints, device=dev)
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2, Constants.ny // 2, Constants.nz, device=dev)
pts = torch.from_numpy(u20k_cloud(0)).to(dev)
yc, yr = synthetic_targets(0, net.Ho, net.Wo)
yc, yr = torch.from_numpy(yc).to(dev), torch.from_numpy(yr).to(dev)
for _ in range(3):
    net.train_step(vox(pts), yc, yr)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    net.train_step(vox(pts), yc, yr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step {1e3*(t1-t0)/K:.2f} ms; total per step {1e3*(t2-t0)/K:.2f} ms")
# phases
def timed(fn):
    torch.cuda.synchronize(); a = time.perf_counter(); fn(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    return 1e3*(b-a), 1e3*(c-a)
s = vox(pts)
print("voxelise  host/total ms", timed(lambda: vox(pts)))
print("forward   host/total ms", timed(lambda: net.forward(s, training=True)))
print("backward  host/total ms", timed(lambda: net.backward(yc, yr)))
print("update    host/total ms", timed(lambda: net.apply_gradients()))

# the recorded form of the same step (what Model.fit and bench.py use on one GPU): one C call per step
from lisec_amd.network import RecordedStep
rec = RecordedStep(net, vox, pts.shape[0])
rec.load(pts, yc, yr)
for _ in range(3):
    rec.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    rec.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"step plan replay ({rec.launches} recorded operations): host enqueue per step {1e3*(t1-t0)/K:.3f} ms; total per step {1e3*(t2-t0)/K:.2f} ms")
t0 = time.perf_counter()
for _ in range(K):
    rec(pts, yc, yr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"load + replay: host per step {1e3*(t1-t0)/K:.3f} ms; total per step {1e3*(t2-t0)/K:.2f} ms")
