import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lisec_amd import ops
from lisec_amd.network import LisecNet
from lisec_amd.params import ParamStore
from lisec_amd.voxelizer import Voxelizer
from oracle import model_ref as M
LYFT = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=100, maxVoxelY=200, maxVoxelZ=8)
rng = np.random.default_rng(5)
n = 20000
pts = np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1).astype(np.float32)
op = M.glorot_params(seed=77, randomize_bn=True)
dev = torch.device("cuda")
net = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
sample = Voxelizer(**LYFT)(pts)
y_cls = torch.from_numpy(rng.integers(0, 3, (100, 200, 2)).astype(np.float32)).to(dev)
y_reg = torch.from_numpy(rng.normal(0, 1, (100, 200, 14)).astype(np.float32)).to(dev)
net.forward(sample, training=True)
net.backward(y_cls, y_reg)
torch.cuda.synchronize()
c = net.layers[0]["conv"]
sparse = net.params.grad_view(net.grad, c.wname).clone().reshape(27, 64, 64)
dz = net.dact["mid1.z"]
ws = torch.zeros(ops.wgrad_workspace_bytes(c.g), dtype=torch.uint8, device=dev)
dense = torch.empty(27, 64, 64, device=dev)
ops.conv_wgrad(c.g, net.dense_grid(), dz, dense, ws)
torch.cuda.synchronize()
# fp64 reference for a few taps on CPU
grid = net.dense_grid().double().cpu(); dzc = dz.double().cpu()
err = (sparse - dense).abs().reshape(27, -1).max(1).values.cpu().numpy()
print("max |dense|", dense.abs().max().item(), "max err sparse-dense per tap:", np.round(err / dense.abs().max().item(), 5))
import torch.nn.functional as F
# exact fp64 for tap (1,1,1) = index 13 and tap 0
def ref_tap(kd, kh, kw):
    # dW[c][n] = sum_m grid[2d-1+kd, h-1+kh, w-1+kw][c] * dz[m][n]
    D, H, W = 8, 200, 400
    out = torch.zeros(64, 64, dtype=torch.float64)
    for d in range(4):
        sd = 2 * d - 1 + kd
        if sd < 0 or sd >= D: continue
        h0, h1 = max(0, 1 - kh), min(H, H + 1 - kh)
        w0, w1 = max(0, 1 - kw), min(W, W + 1 - kw)
        g = grid[sd, h0 - 1 + kh:h1 - 1 + kh, w0 - 1 + kw:w1 - 1 + kw].reshape(-1, 64)
        z = dzc[d, h0:h1, w0:w1].reshape(-1, 64)
        out += g.T @ z
    return out
for tap in (0, 13, 26, 4):
    kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
    r = ref_tap(kd, kh, kw)
    print("tap", tap, "sparse err", ((sparse[tap].double().cpu() - r).abs().max() / r.abs().max()).item(),
          "dense err", ((dense[tap].double().cpu() - r).abs().max() / r.abs().max()).item())
