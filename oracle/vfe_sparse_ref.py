"""Sparse-exact restatement of the dense VFE stack -- TEST INFRASTRUCTURE ONLY.

Only tests/ (and smoke/bench's checker legs) may import this module.

The reference feeds a dense (D,H,W,T,6) tensor through Dense(no bias)->BN->ReLU->max/repeat/
concat (model_training.py:155-186, :231-235).  Because Dense has no bias and there is no
point mask, every dense row is one of
    R  a real point row                              (weight 1)
    P  the zero pad row of a non-empty voxel          (weight T - s_v)
    E  the zero row of an empty voxel                 (one class, weight T * n_empty)
and rows of one class are bit-identical through the whole stack.  This module evaluates
one representative per class with its multiplicity (forward AND backward) in float64 numpy
and is proven equal to oracle/model_ref.py (dense torch autograd) by
tests/test_oracle_model.py.  It is the derivation the HIP VFE kernels follow
(lisec_amd/csrc/vfe.hip); "parity unpinned" status is inherited from model_ref.
"""
import numpy as np

EPS = 1e-3


def build_rows(feats, npts, T, ncells):
    """feats (V,T,6), npts (V,) -> class rows.

    Returns x (R,6), w (R,) dense-row multiplicity, vox (R,) voxel id (V = the virtual empty
    voxel), seg (V+2,) row offsets per voxel.
    """
    V = len(npts)
    xs, ws, vs, seg = [], [], [], [0]
    for v in range(V):
        s = int(npts[v])
        xs.append(feats[v, :s].astype(np.float64))
        ws.extend([1.0] * s)
        vs.extend([v] * s)
        if s < T:
            xs.append(np.zeros((1, 6)))
            ws.append(float(T - s))
            vs.append(v)
        seg.append(len(ws))
    n_empty = ncells - V
    if n_empty > 0:
        xs.append(np.zeros((1, 6)))
        ws.append(float(T) * n_empty)
        vs.append(V)
    seg.append(len(ws))
    x = np.concatenate(xs) if xs else np.zeros((0, 6))
    return x, np.array(ws), np.array(vs, dtype=np.int64), np.array(seg, dtype=np.int64)


def _seg_max(a, seg):
    nv = len(seg) - 1
    out = np.zeros((nv, a.shape[1]))
    arg = np.zeros((nv, a.shape[1]), dtype=np.int64)
    for v in range(nv):
        lo, hi = seg[v], seg[v + 1]
        if hi > lo:
            arg[v] = lo + np.argmax(a[lo:hi], axis=0)      # first maximum wins
            out[v] = a[arg[v], np.arange(a.shape[1])]
    return out, arg


def _bn_fwd(y, w, N, gamma, beta, training, mm, mv):
    if training:
        mean = (w[:, None] * y).sum(0) / N
        var = (w[:, None] * (y - mean) ** 2).sum(0) / N
    else:
        mean, var = mm, mv
    inv = gamma / np.sqrt(var + EPS)
    z = y * inv + (beta - mean * inv)
    return z, mean, var, inv


def forward(p, x, w, vox, seg, N, training=True):
    """p: dict of float64 numpy params (names of oracle/model_ref.param_specs).
    Returns (out (nvox,64) per voxel incl. virtual, cache)."""
    cache = dict(x=x, w=w, vox=vox, seg=seg, N=N)
    h = x
    for li, name in enumerate(("vfe1", "vfe2", "fcn")):
        W = p[f"{name}.dense.kernel"]
        y = h @ W
        z, mean, var, inv = _bn_fwd(y, w, N, p[f"{name}.bn.gamma"], p[f"{name}.bn.beta"], training,
                                    p[f"{name}.bn.moving_mean"], p[f"{name}.bn.moving_variance"])
        a = np.maximum(z, 0.0)
        pool, arg = _seg_max(a, seg)
        cache[name] = dict(h=h, y=y, z=z, mean=mean, var=var, inv=inv, a=a, arg=arg)
        if li < 2:
            h = np.concatenate([pool[vox], a], axis=1)     # pooled half first (:164-165)
    return pool, cache


def backward(p, cache, dout):
    """dout (nvox,64): gradient wrt the per-voxel output, SUMMED over the identical copies of a
    class (the virtual voxel gets the sum of the grid gradient over all empty cells).
    Returns dict of parameter gradients (training-mode BN)."""
    w, vox, seg, N = cache["w"], cache["vox"], cache["seg"], cache["N"]
    grads = {}
    R = len(w)
    GA = np.zeros((R, 64))
    arg = cache["fcn"]["arg"]
    for c in range(64):
        np.add.at(GA[:, c], arg[:, c], dout[:, c])
    for name in ("fcn", "vfe2", "vfe1"):
        c_ = cache[name]
        W = p[f"{name}.dense.kernel"]
        GZ = GA * (c_["z"] > 0)
        sigma_inv = 1.0 / np.sqrt(c_["var"] + EPS)
        yhat = (c_["y"] - c_["mean"]) * sigma_inv
        dbeta = GZ.sum(0)
        dgamma = (GZ * yhat).sum(0)
        GY = c_["inv"] * (GZ - w[:, None] * dbeta / N - w[:, None] * yhat * dgamma / N)
        grads[f"{name}.bn.beta"] = dbeta
        grads[f"{name}.bn.gamma"] = dgamma
        grads[f"{name}.dense.kernel"] = c_["h"].T @ GY
        if name == "vfe1":
            break
        GH = GY @ W.T
        half = W.shape[0] // 2
        lower = "vfe2" if name == "fcn" else "vfe1"
        GA = GH[:, half:].copy()
        nv = len(seg) - 1
        gpool = np.zeros((nv, half))
        np.add.at(gpool, vox, GH[:, :half])
        larg = cache[lower]["arg"]
        for c in range(half):
            np.add.at(GA[:, c], larg[:, c], gpool[:, c])
    return grads
