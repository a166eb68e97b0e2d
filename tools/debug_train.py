import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from lisec_amd.network import LisecNet
from lisec_amd.params import ParamStore
from lisec_amd.voxelizer import Voxelizer
from oracle import model_ref as M
from oracle import voxel_ref
from test_gpu_network import SMALL, small_cloud

op = M.glorot_params(seed=33, randomize_bn=True)
dev = torch.device("cuda")
net = LisecNet(16, 32, 8, 35, params=ParamStore(dev, init=op))
vox = Voxelizer(**SMALL)
rng = np.random.default_rng(8)
p64 = {k: v.double() for k, v in op.items()}
vel = {n: torch.zeros_like(p64[n]) for n, _, k in M.param_specs() if M.is_trainable(k)}
shape = (8, 16, 32, 35, 6)
for it in range(3):
    pts = small_cloud(seed=40 + it)
    y_cls = rng.integers(0, 3, (8, 16, 2)).astype(np.float32)
    y_reg = rng.normal(0, 1, (8, 16, 14)).astype(np.float32)
    ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **SMALL)
    dense = torch.from_numpy(voxel_ref.to_dense(ref_vox, shape))[None].double()
    p64 = {k: v.float().double() for k, v in p64.items()}
    vel = {k: v.float().double() for k, v in vel.items()}
    loss_r, grads_r, p64_new, vel_new, _ = M.train_step(p64, vel, dense, torch.from_numpy(y_cls)[None].double(),
                                                    torch.from_numpy(y_reg)[None].double(), it)
    # re-synchronise with the oracle's state so that rounding differences do not compound
    net.params.load_dict({k: v.float() for k, v in p64.items()})
    net._prepare_training()
    for n_, v_ in vel.items():
        net.params.grad_view(net.velocity, n_).copy_(v_.float())
    net.iterations = it
    sample = vox(pts)
    lo = net.train_step(sample, torch.from_numpy(y_cls).to(dev), torch.from_numpy(y_reg).to(dev))
    torch.cuda.synchronize()
    print("step", it, "loss", lo[0].item(), loss_r.item())
    for name, g in grads_r.items():
        got = net.params.grad_view(net.grad, name).cpu().numpy()
        ref = g.numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        if err > 1e-3:
            print(f"   {name:28s} relerr {err:.3e} maxref {np.abs(ref).max():.3e}")
    got_p = net.params.to_dict()
    worst = max(np.abs(got_p[k] - v.numpy()).max() / (np.abs(v.numpy()).max() + 1e-30) for k, v in p64_new.items())
    print("   worst relative param error after update", worst)
    p64, vel = p64_new, vel_new
