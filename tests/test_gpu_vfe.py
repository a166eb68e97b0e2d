"""HIP sparse-exact VFE (C ABI) vs the dense torch oracle (small grid) and the fp64 sparse oracle
(full Lyft grid).  Tolerance: rtol 1e-3 with atol 1e-3*max|ref| (BASELINE north_star); observed
errors are ~1e-6 because both sides are fp32 with fp64 statistics."""
import numpy as np
import pytest
import torch

from conftest import LYFT

pytestmark = pytest.mark.gpu


def _oracle_params(seed):
    from oracle import model_ref as M
    return M.glorot_params(seed=seed, randomize_bn=True)


def _close(got, ref, rtol=1e-3):
    ref = np.asarray(ref, dtype=np.float64)
    atol = 1e-3 * np.abs(ref).max()
    err = np.abs(np.asarray(got, dtype=np.float64) - ref)
    assert (err <= atol + rtol * np.abs(ref)).all(), f"max err {err.max():.3e} (atol {atol:.3e})"
    return err.max() / max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("training", [True, False])
def test_vfe_small_grid_vs_dense_oracle(training):
    from lisec_amd.params import ParamStore
    from lisec_amd.vfe import VFEStack
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    from oracle import voxel_ref

    cfg = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)
    rng = np.random.default_rng(1)
    n = 3000       # dense enough that several voxels hold > 35 points and many are full
    pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1)
    pts[:600, :2] *= 0.1
    pts[:600, 2] = 0.5 + 0.5 * rng.uniform(0, 1, 600)
    pts = pts.astype(np.float32)
    op = _oracle_params(5)
    dev = torch.device("cuda")
    store = ParamStore(dev, init=op)
    sample = Voxelizer(**cfg)(pts)
    grid = VFEStack(store).forward(sample, training=training).cpu().numpy()

    ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **cfg)
    assert ref_vox["counts"].max() > 35
    dense = torch.from_numpy(voxel_ref.to_dense(ref_vox, (8, 16, 32, 35, 6)))[None]
    p64 = {k: v.double() for k, v in op.items()}
    stats = {}
    h = M._vfe(dense.double(), p64, "vfe1", training, stats)
    h = M._vfe(h, p64, "vfe2", training, stats)
    h = M._fcn(h, p64, "fcn", training, stats)
    ref = h.max(dim=-2).values[0].numpy()
    rel = _close(grid, ref)
    assert rel < 1e-4
    if training:
        new = M.updated_moving_stats(p64, stats)
        got = store.to_dict()
        for k, v in new.items():
            _close(got[k], v.numpy(), rtol=1e-5)


def test_vfe_full_lyft_grid_vs_sparse_oracle():
    from lisec_amd.params import ParamStore
    from lisec_amd.vfe import VFEStack
    from lisec_amd.voxelizer import Voxelizer
    from oracle import vfe_sparse_ref as S
    from oracle import voxel_ref

    rng = np.random.default_rng(0)
    n = 20000
    pts = np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1).astype(np.float32)
    op = _oracle_params(9)
    store = ParamStore(torch.device("cuda"), init=op)
    sample = Voxelizer(**LYFT)(pts)
    grid = VFEStack(store).forward(sample, training=True)
    torch.cuda.synchronize()
    ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **LYFT)
    ncells = 8 * 200 * 400
    x, w, vox, seg = S.build_rows(ref_vox["feats"], ref_vox["npts"], 35, ncells)
    pn = {k: v.double().numpy() for k, v in op.items()}
    out, _ = S.forward(pn, x, w, vox, seg, N=float(ncells * 35), training=True)
    g = grid.cpu().numpy().reshape(ncells, 64)
    c = ref_vox["coords"]
    cells = (c[:, 0] * 200 + c[:, 1]) * 400 + c[:, 2]
    _close(g[cells], out[:-1])
    empty = np.ones(ncells, bool)
    empty[cells] = False
    # every empty cell holds the same (non-zero) constant
    const = g[empty][0]
    assert (g[empty] == const[None, :]).all()
    _close(const, out[-1])
    assert np.abs(const).max() > 0


@pytest.mark.parametrize("path", ["tiled", "valu"])
@pytest.mark.parametrize("grid", ["small", "lyft"])
def test_vfe_backward_vs_sparse_oracle(grid, path):
    """Gradients of the VFE variables from a random grid gradient, vs the fp64 row-class oracle
    (itself proven equal to dense torch autograd in tests/test_oracle_model.py).  Both backward paths: layers 3 and 2
    on 32-row MFMA tiles with the forward's saved winner slots (the default), and row by row per voxel."""
    from lisec_amd.params import ParamStore
    from lisec_amd.vfe import VFEStack
    from lisec_amd.voxelizer import Voxelizer
    from oracle import vfe_sparse_ref as S
    from oracle import voxel_ref

    rng = np.random.default_rng(4)
    if grid == "small":
        cfg = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)
        n = 3000
        pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1)
        pts[:600, :2] *= 0.1
        pts[:600, 2] = 0.5 + 0.5 * rng.uniform(0, 1, 600)
    else:
        cfg = LYFT
        n = 20000
        pts = np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1)
    pts = pts.astype(np.float32)
    D, H, W = cfg["maxVoxelZ"], 2 * cfg["maxVoxelX"], 2 * cfg["maxVoxelY"]
    ncells = D * H * W
    op = _oracle_params(13)
    dev = torch.device("cuda")
    store = ParamStore(dev, init=op)
    vfe = VFEStack(store)
    vfe.tiled, vfe.tiled_min_points = path == "tiled", 0
    sample = Voxelizer(**cfg)(pts)
    vfe.forward(sample, training=True)
    dgrid = torch.randn(D, H, W, 64, device=dev) * (1.0 / ncells) ** 0.5
    grad = torch.zeros_like(store.theta)
    vfe.backward(dgrid, grad)
    torch.cuda.synchronize()

    ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **cfg)
    x, w, vox, seg = S.build_rows(ref_vox["feats"], ref_vox["npts"], 35, ncells)
    pn = {k: v.double().numpy() for k, v in op.items()}
    _, cache = S.forward(pn, x, w, vox, seg, N=float(ncells * 35), training=True)
    dg = dgrid.cpu().numpy().astype(np.float64).reshape(ncells, 64)
    c = ref_vox["coords"]
    cells = (c[:, 0] * H + c[:, 1]) * W + c[:, 2]
    empty = np.ones(ncells, bool)
    empty[cells] = False
    dout = np.concatenate([dg[cells], dg[empty].sum(0, keepdims=True)])
    ref = S.backward(pn, cache, dout)
    for name, r in ref.items():
        got = store.grad_view(grad, name).cpu().numpy()
        _close(got, r, rtol=2e-3)


def _decode_row_stats(words):
    """int64 row_stats -> the 27 moments (sum over replicas of hi * 2^-8 + lo * 2^-40)."""
    from lisec_amd import _lib
    w = np.asarray(words[:_lib.ROW_STATS_MOMENT_WORDS], dtype=np.int64).reshape(_lib.ROW_STATS_REPLICAS, 27, 2)
    return w[:, :, 0].sum(0) / 256.0 + w[:, :, 1].sum(0) / 1099511627776.0


def test_voxeliser_row_moments_and_vfe_without_them():
    """lisec_voxelize's side output row_stats: the 6 first and 21 second moments of the feature rows it wrote (what
    the closed-form statistics of the VFE's first BatchNormalization are made of), and the scratch behind them left
    zeroed by a training forward.  A caller without row_stats (NULL) gets the same grid: lisec_vfe_forward then sums
    the moments itself."""
    from lisec_amd import _lib
    from lisec_amd.params import ParamStore
    from lisec_amd.vfe import VFEStack
    from lisec_amd.voxelizer import Voxelizer, host_row_stats
    rng = np.random.default_rng(11)
    n = 20000
    pts = np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1).astype(np.float32)
    sample = Voxelizer(**LYFT)(pts)
    h = sample.to_host()
    rows = h["rows"].astype(np.float64)
    want = np.array([rows[:, j].sum() for j in range(6)] +
                    [(rows[:, j] * rows[:, k]).sum() for j in range(6) for k in range(j, 6)])
    got = _decode_row_stats(sample.row_stats.cpu().numpy())
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    assert np.allclose(_decode_row_stats(host_row_stats(h["rows"])), want, rtol=1e-12, atol=1e-9)
    assert (sample.row_stats.cpu().numpy()[_lib.ROW_STATS_MOMENT_WORDS:] == 0).all()
    op = _oracle_params(9)
    dev = torch.device("cuda")
    a = VFEStack(ParamStore(dev, init=op))
    g1 = a.forward(sample, training=True).cpu().numpy()
    state1 = a.params.state.cpu().numpy().copy()
    assert (sample.row_stats.cpu().numpy()[_lib.ROW_STATS_MOMENT_WORDS:] == 0).all()        # scratch re-zeroed
    g1b = a.forward(sample, training=True).cpu().numpy()                                    # same sample again
    assert np.array_equal(g1, g1b)
    keep, sample.row_stats = sample.row_stats, None
    b = VFEStack(ParamStore(dev, init=op))
    g2 = b.forward(sample, training=True).cpu().numpy()
    sample.row_stats = keep
    assert np.allclose(g1, g2, rtol=1e-6, atol=1e-7)
    assert np.allclose(state1, b.params.state.cpu().numpy(), rtol=1e-6, atol=1e-9)      # moving statistics too
