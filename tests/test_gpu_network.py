"""End-to-end forward (voxeliser -> VFE -> 3D-conv middle -> RPN heads) through the C ABI vs the
dense oracle (oracle/model_ref.py) on identical voxel grids.
Tolerance: BASELINE north_star -- RPN outputs rtol 1e-3, paired with atol 1e-3*max|ref|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)


def small_cloud(seed=1, n=3000):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1)
    pts[:600, :2] *= 0.1
    pts[:600, 2] = 0.5 + 0.5 * rng.uniform(0, 1, 600)
    return pts.astype(np.float32)


def close(got, ref, rtol=1e-3, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    atol = rtol * np.abs(ref).max()
    err = np.abs(got - ref)
    assert (err <= atol + rtol * np.abs(ref)).all(), f"{what}: max err {err.max():.3e}, atol {atol:.3e}"
    return err.max() / max(np.abs(ref).max(), 1e-30)


def oracle_forward(op, pts, cfg, training, dtype=torch.float64):
    from oracle import model_ref as M
    from oracle import voxel_ref
    vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **cfg)
    shape = (cfg["maxVoxelZ"], 2 * cfg["maxVoxelX"], 2 * cfg["maxVoxelY"], cfg["sampleSize"], 6)
    dense = torch.from_numpy(voxel_ref.to_dense(vox, shape))[None].to(dtype)
    p = {k: v.to(dtype) for k, v in op.items()}
    taps, stats = {}, {}
    cls, reg = M.forward(p, dense, training=training, stats=stats, taps=taps)
    return cls, reg, taps, stats, p


@pytest.mark.parametrize("training", [False, True])
def test_forward_small_grid_vs_dense_oracle(training):
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    op = M.glorot_params(seed=21, randomize_bn=True)
    pts = small_cloud()
    cls_r, reg_r, taps, stats, p64 = oracle_forward(op, pts, SMALL, training)
    dev = torch.device("cuda")
    net = LisecNet(16, 32, 8, 35, params=ParamStore(dev, init=op))
    sample = Voxelizer(**SMALL)(pts)
    cls, reg = net.forward(sample, training=training)
    torch.cuda.synchronize()
    a = net.act
    close(a["grid"].cpu().numpy(), taps["vfe_grid"][0].numpy(), what="vfe grid")
    for i in (1, 2, 3):
        close(a[f"mid{i}.u"].cpu().numpy(), taps[f"mid{i}"][0].numpy(), what=f"mid{i}")
    close(a["concat"].cpu().numpy(), taps["concat"][0].numpy(), what="concat")
    close(cls.cpu().numpy(), cls_r.numpy(), what="cls")
    close(reg.cpu().numpy(), reg_r.numpy(), what="reg")
    assert cls.shape == (1, 8, 16, 2) and reg.shape == (1, 8, 16, 14)
    if training:
        new = M.updated_moving_stats(p64, stats)
        got = net.params.to_dict()
        for k, v in new.items():
            close(got[k], v.numpy(), rtol=1e-4, what=k)
