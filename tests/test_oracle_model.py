"""Oracle self-consistency: known-answer shapes/param counts of the dense restatement, and the
sparse-exact (row class) VFE derivation == dense autograd, forward and backward."""
import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import vfe_sparse_ref as S


def test_param_count_matches_reference_graph():
    # 6 491 024 trainable parameters: SURVEY section 2 table (derived from model_training.py:222-257)
    n = sum(int(np.prod(s)) for _, s, k in M.param_specs() if M.is_trainable(k))
    assert n == 6_491_024


def test_known_answer_shapes():
    # model.png (conv3d onward) and rpnToRegion.py:116-117: (100,200,2)/(100,200,14) at the Lyft grid;
    # here on an 8x-smaller H,W: every spatial shape scales, channel counts do not.
    p = M.glorot_params()
    x = torch.zeros(1, 8, 24, 48, 35, 6)
    taps = {}
    cls, reg = M.forward(p, x, training=False, taps=taps)
    assert tuple(cls.shape) == (1, 12, 24, 2) and tuple(reg.shape) == (1, 12, 24, 14)
    assert tuple(taps["mid1"].shape) == (1, 4, 24, 48, 64)
    assert tuple(taps["mid2"].shape) == (1, 2, 24, 48, 64)
    assert tuple(taps["mid3"].shape) == (1, 1, 24, 48, 64)
    assert tuple(taps["rpn1"].shape) == (1, 12, 24, 128)
    assert tuple(taps["rpn2"].shape) == (1, 6, 12, 128)
    assert tuple(taps["rpn3"].shape) == (1, 3, 6, 256)
    assert tuple(taps["concat"].shape) == (1, 12, 24, 768)


def _random_voxels(rng, D, H, W, T, nvox):
    cells = rng.choice(D * H * W, nvox, replace=False)
    cells.sort()
    npts = rng.integers(1, T + 1, nvox)
    npts[0] = T                                   # a full voxel (no pad row)
    feats = np.zeros((nvox, T, 6), np.float64)
    for v in range(nvox):
        feats[v, :npts[v]] = rng.normal(0, 2.0, (npts[v], 6))
    dense = np.zeros((D * H * W, T, 6))
    dense[cells] = feats
    return cells, npts, feats, dense.reshape(1, D, H, W, T, 6)


@pytest.mark.parametrize("training", [True, False])
def test_sparse_exact_vfe_equals_dense(training):
    rng = np.random.default_rng(0)
    D, H, W, T = 4, 6, 8, 5
    cells, npts, feats, dense = _random_voxels(rng, D, H, W, T, 23)
    p64 = {k: v.double() for k, v in M.glorot_params(seed=3, randomize_bn=True).items()}
    pn = {k: v.numpy() for k, v in p64.items()}
    names = [n for n, _, k in M.param_specs() if n.split(".")[0] in ("vfe1", "vfe2", "fcn") and M.is_trainable(k)]
    for n in names:
        p64[n].requires_grad_(True)

    # dense: model_ref VFE part only
    x = torch.tensor(dense)
    h = M._vfe(x, p64, "vfe1", training, None)
    h = M._vfe(h, p64, "vfe2", training, None)
    h = M._fcn(h, p64, "fcn", training, None)
    grid = h.max(dim=-2).values                                  # (1,D,H,W,64)
    dgrid = torch.tensor(rng.normal(size=grid.shape))
    (grid * dgrid).sum().backward()

    ncells = D * H * W
    xr, w, vox, seg = S.build_rows(feats, npts, T, ncells)
    out, cache = S.forward(pn, xr, w, vox, seg, N=float(ncells * T), training=training)
    g = grid.detach().numpy().reshape(ncells, 64)
    empty = np.ones(ncells, bool)
    empty[cells] = False
    assert np.allclose(g[cells], out[:-1], rtol=1e-10, atol=1e-12)
    assert np.allclose(g[empty], out[-1][None, :], rtol=1e-10, atol=1e-12)
    if not training:
        return
    dg = dgrid.numpy().reshape(ncells, 64)
    dout = np.concatenate([dg[cells], dg[empty].sum(0, keepdims=True)])
    grads = S.backward(pn, cache, dout)
    for n in names:
        ref = p64[n].grad.numpy()
        assert np.allclose(grads[n], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max())), n
