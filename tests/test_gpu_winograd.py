"""Winograd F(2x2, 3x3) form of the stride-1 3x3 contractions (lisec_conv_forward_winograd, csrc/wino.hip) against the fp64
restatement oracle/conv_ref.py -- the same oracle, inputs and bound (relative L2 <= 2e-6) as the direct kernels in
tests/test_gpu_lyft_layers.py, so the two forms are held to one standard: small maps that exercise every edge (odd maps,
partial 8 x 8 tile blocks, several column blocks, depth strides and paddings in both gather modes, BatchNormalization + ReLU
on load, accumulate, gate, both sinks) and the Lyft geometries the training step runs it on.
Reference statements: model_training.py:193 (Conv3D), :203 (Conv2D), :204-206 (BN + ReLU), :299 (fit: the data gradients)."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-6
H, W = 200, 400


def rel_l2(got, ref):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))


def make_bn(rng, c, dev):
    scale, shift = rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c)
    mean, invstd = rng.normal(0, 0.2, c), rng.uniform(0.7, 1.4, c)
    st = np.concatenate([scale, shift, mean, invstd]).astype(np.float32)
    return torch.from_numpy(st).to(dev), (st[:c].astype(np.float64), st[c:2 * c].astype(np.float64))


# name, mode, in dims, out dims, KD, depth stride, depth pad, cin, cout, BN+ReLU on load
SMALL = [
    ("2d 16x16", 0, (1, 16, 16), (1, 16, 16), 1, 1, 0, 16, 64, False),
    ("2d odd map 25x50, 2 column blocks, bn", 0, (1, 25, 50), (1, 25, 50), 1, 1, 0, 16, 128, True),
    ("2d 18x34 (partial tile blocks), cout 40", 0, (1, 18, 34), (1, 18, 34), 1, 1, 0, 48, 40, True),
    ("3d valid depth (mid2 shape)", 0, (4, 12, 20), (2, 12, 20), 3, 1, 0, 16, 64, False),
    ("3d depth stride 2 pad 1 (mid3 shape)", 0, (2, 10, 18), (1, 10, 18), 3, 2, 1, 16, 64, False),
    ("3d depth stride 2 pad 1, 8 planes", 0, (8, 6, 10), (4, 6, 10), 3, 2, 1, 32, 64, True),
    ("transposed 2d", 1, (1, 14, 22), (1, 14, 22), 1, 1, 0, 16, 64, False),
    ("transposed 3d valid depth", 1, (2, 12, 20), (4, 12, 20), 3, 1, 0, 16, 64, False),
    ("transposed 3d depth stride 2 pad 1", 1, (1, 10, 18), (2, 10, 18), 3, 2, 1, 16, 64, False),
    ("transposed 3d depth stride 2 pad 1, 4 planes", 1, (4, 6, 10), (8, 6, 10), 3, 2, 1, 16, 72, False),
]


def run_case(name, mode, ind, outd, KD, sd, pd, cin, cout, xf, rng, dev, extra=None):
    from lisec_amd import ops
    from oracle import conv_ref
    k, s, p = (KD, 3, 3), (sd, 1, 1), (pd, 1, 1)
    ntaps = KD * 9
    x = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
    Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    bn_dev, bn_ref = make_bn(rng, cin, dev) if xf else (None, None)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    wd = torch.from_numpy(Wt).to(dev)
    wu = ops.pack_weights_winograd(wd, KD, cin, cout, cin * cout, cout, 1, flip=(mode == 1))
    ref = conv_ref.conv_forward(x, Wt, outd, k, s, p, mode=mode, bias=b, in_bn=bn_ref, relu=xf)
    return g, x, Wt, b, bn_dev, wu, ref


@pytest.mark.parametrize("case", SMALL, ids=[c[0] for c in SMALL])
def test_small_maps_against_the_oracle(case):
    from lisec_amd import ops
    name, mode, ind, outd, KD, sd, pd, cin, cout, xf = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    g, x, Wt, b, bn_dev, wu, ref = run_case(*case, rng, dev)
    flags = ops.IN_RELU if xf else 0
    assert ops.winograd_supported(g, in_bn=xf, flags=flags)
    out = torch.full((*outd, cout), float("nan"), device=dev)
    ops.conv_forward_winograd(g, torch.from_numpy(x).to(dev), wu, out, bias=torch.from_numpy(b).to(dev), in_bn=bn_dev,
                              flags=flags)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all(), f"{name}: positions left unwritten"
    e = rel_l2(got, ref)
    assert e <= TOL, f"{name}: relative L2 {e:.2e}"
    # the direct form on the same operands: both within the bound of the same oracle
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), KD * 9, cin, cout, cin * cout, cout, 1)
    out2 = torch.empty_like(out)
    ops.conv_forward(g, torch.from_numpy(x).to(dev), wp, out2, bias=torch.from_numpy(b).to(dev), in_bn=bn_dev, flags=flags)
    torch.cuda.synchronize()
    assert rel_l2(out2.cpu().numpy(), ref) <= TOL


def test_accumulate_gate_and_relu():
    """LISEC_CONV_ACCUMULATE onto an existing gradient, the out_mask gate and LISEC_CONV_OUT_RELU, as the direct form
    applies them: (+ bias, + previous, gate, ReLU) in that order."""
    from lisec_amd import ops
    dev = torch.device("cuda")
    rng = np.random.default_rng(11)
    case = ("acc", 1, (1, 20, 36), (1, 20, 36), 1, 1, 0, 16, 64, False)
    g, x, Wt, b, _, wu, ref = run_case(*case, rng, dev)
    prev = rng.normal(0, 1, ref.shape).astype(np.float32)
    act = rng.normal(0, 1, ref.shape).astype(np.float32)
    out = torch.from_numpy(prev.copy()).to(dev)
    ops.conv_forward_winograd(g, torch.from_numpy(x).to(dev), wu, out, bias=torch.from_numpy(b).to(dev),
                              flags=ops.ACCUMULATE | ops.OUT_RELU, out_mask=torch.from_numpy(act).to(dev))
    torch.cuda.synchronize()
    want = np.maximum(np.where(act > 0, ref + prev, 0.0), 0.0)
    assert rel_l2(out.cpu().numpy(), want) <= TOL


def test_unsupported_arguments_are_refused():
    from lisec_amd import _lib, ops
    dev = torch.device("cuda")
    g = ops.geom(0, (1, 16, 16), (1, 8, 8), (1, 3, 3), (1, 2, 2), (0, 1, 1), 16, 64)          # stride 2
    assert not ops.winograd_supported(g)
    g = ops.geom(0, (1, 16, 16), (1, 16, 16), (1, 3, 3), (1, 1, 1), (0, 1, 1), 24, 64)        # Cin % 16
    assert not ops.winograd_supported(g)
    x = torch.zeros(1, 16, 16, 24, device=dev)
    with pytest.raises(_lib.LisecError):
        ops.conv_forward_winograd(g, x, torch.zeros(16 * 64 * 8 * 3, device=dev), torch.zeros(1, 16, 16, 64, device=dev))


# the layers the training step runs in this form, at the Lyft grid: name, mode, in dims, out dims, KD, sd, pd, cin, cout, kind
LYFT = [
    ("mid2 forward", 0, (4, H, W), (2, H, W), 3, 1, 0, 64, 64, "fwd"),
    ("mid3 forward", 0, (2, H, W), (1, H, W), 3, 2, 1, 64, 64, "fwd"),
    ("rpn1.conv1 forward", 0, (1, 100, 200), (1, 100, 200), 1, 1, 0, 128, 128, "fwd_bn"),
    ("mid2 data gradient", 1, (2, H, W), (4, H, W), 3, 1, 0, 64, 64, "mask"),
    ("mid3 data gradient", 1, (1, H, W), (2, H, W), 3, 2, 1, 64, 64, "mask"),
    ("rpn1.conv1 data gradient", 1, (1, 100, 200), (1, 100, 200), 1, 1, 0, 128, 128, "bn"),
]


@pytest.mark.parametrize("case", LYFT, ids=[c[0] for c in LYFT])
def test_lyft_geometries(case):
    """Forward with the batch statistics summed into a forward sink (bnstate finalised inside the call); data gradients
    with the Dense gate ('mask') or the statistics of the BatchNormalization backward it is about to cross ('bn')."""
    from lisec_amd import ops
    name, mode, ind, outd, KD, sd, pd, cin, cout, kind = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    xf = kind == "fwd_bn"
    g, x, Wt, b, bn_dev, wu, ref = run_case(name, mode, ind, outd, KD, sd, pd, cin, cout, xf, rng, dev)
    M = outd[0] * outd[1] * outd[2]
    out = torch.full((*outd, cout), float("nan"), device=dev)
    xd = torch.from_numpy(x).to(dev)
    if kind in ("fwd", "fwd_bn"):
        gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
        beta = torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32)).to(dev)
        bnstate = torch.zeros(4 * cout, device=dev)
        sink = ops.BnSink(cout, M, dev, gamma=gamma, beta=beta, bnstate=bnstate)
        flags = ops.IN_RELU if xf else 0
        assert ops.winograd_supported(g, in_bn=xf, flags=flags, sink=sink)
        ops.conv_forward_winograd(g, xd, wu, out, bias=torch.from_numpy(b).to(dev), in_bn=bn_dev, flags=flags, sink=sink)
        torch.cuda.synchronize()
        e = rel_l2(out.cpu().numpy(), ref)
        assert e <= TOL, f"{name}: relative L2 {e:.2e}"
        y = ref.reshape(M, cout)
        mean, var = y.mean(0), y.var(0)
        inv = 1.0 / np.sqrt(var + 1e-3)
        st = bnstate.cpu().numpy().astype(np.float64)
        gm, bt = gamma.cpu().numpy().astype(np.float64), beta.cpu().numpy().astype(np.float64)
        assert np.allclose(st[2 * cout:3 * cout], mean, rtol=0, atol=2e-6 * np.abs(y).max()), f"{name}: batch mean"
        assert np.allclose(st[3 * cout:], inv, rtol=2e-5), f"{name}: inverse std"
        assert np.allclose(st[:cout], gm * inv, rtol=2e-5) and np.allclose(st[cout:2 * cout], bt - mean * gm * inv,
                                                                             rtol=1e-4, atol=1e-5), f"{name}: scale / shift"
        return
    ref = ref - b.astype(np.float64)                 # data gradients carry no bias
    if kind == "mask":
        act = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
        ops.conv_forward_winograd(g, xd, wu, out, out_mask=torch.from_numpy(act).to(dev))
        torch.cuda.synchronize()
        e = rel_l2(out.cpu().numpy(), np.where(act > 0, ref, 0.0))
        assert e <= TOL, f"{name}: relative L2 {e:.2e}"
        return
    y = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
    st_dev, (scale, shift) = make_bn(rng, cout, dev)
    st = st_dev.cpu().numpy().astype(np.float64)
    dgamma, dbeta = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
    sink = ops.BnSink(cout, M, dev, dgamma=dgamma, dbeta=dbeta)
    bwd = (torch.from_numpy(y).to(dev), st_dev, True)
    assert ops.winograd_supported(g, bwd=bwd, sink=sink)
    ops.conv_forward_winograd(g, xd, wu, out, bwd=bwd, sink=sink)
    torch.cuda.synchronize()
    e = rel_l2(out.cpu().numpy(), ref)
    assert e <= TOL, f"{name}: relative L2 {e:.2e}"
    y64 = y.astype(np.float64).reshape(M, cout)
    dz = np.where(y64 * scale + shift > 0, ref.reshape(M, cout), 0.0)
    yhat = (y64 - st[2 * cout:3 * cout]) * st[3 * cout:]
    db, dg = dz.sum(0), (dz * yhat).sum(0)
    tol_s = 3e-6 * np.sqrt(M) * np.abs(dz).max()
    assert np.abs(dbeta.cpu().numpy() - db).max() <= tol_s and np.abs(dgamma.cpu().numpy() - dg).max() <= 3 * tol_s, name
    coef = sink.coef.cpu().numpy().astype(np.float64)
    assert np.abs(coef[:cout] - db / M).max() <= tol_s / M and np.abs(coef[cout:] - dg / M).max() <= 3 * tol_s / M, name


@pytest.mark.parametrize("case", [("small", (2, 12, 20), (4, 12, 20), 3, 1, 0), ("mid2", (2, H, W), (4, H, W), 3, 1, 0),
                                  ("mid3", (1, H, W), (2, H, W), 3, 2, 1)], ids=["small", "mid2", "mid3"])
def test_data_gradient_with_dense_tail(case):
    """The Dense(64, relu) data gradient of the block below riding on the block (lisec_conv_extras.tail_w, model_training.py:195
    backwards), as tests/test_gpu_lyft_layers.py holds the direct kernels to it: out = gate(dy (*) W^T), out2 = out @ Wd^T, and
    (sum dz, sum dz*yhat) of out2 for the BatchNormalization under the Dense -- against the two separate steps in fp64."""
    from lisec_amd import ops
    from oracle import conv_ref
    name, ind, outd, KD, sd, pd = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(zlib.crc32(("tail" + name).encode()))
    k, s, p = (KD, 3, 3), (sd, 1, 1), (pd, 1, 1)
    ntaps = KD * 9
    dy = rng.normal(0, 1, (*ind, 64)).astype(np.float32)
    Wt = (rng.normal(0, 1, (ntaps, 64, 64)) / np.sqrt(ntaps * 64)).astype(np.float32)
    Wd = (rng.normal(0, 1, (1, 64, 64)) / 8).astype(np.float32)          # the tail kernel as the contraction sees it: (c, j)
    M = outd[0] * outd[1] * outd[2]
    g = ops.geom(1, ind, outd, k, s, p, 64, 64)
    wu = ops.pack_weights_winograd(torch.from_numpy(Wt).to(dev), KD, 64, 64, 64 * 64, 64, 1, flip=True)
    wdp = ops.pack_weights(torch.from_numpy(Wd).to(dev), 1, 64, 64, 0, 64, 1)
    act = rng.normal(0, 1, (*outd, 64)).astype(np.float32)
    y = rng.normal(0, 1, (*outd, 64)).astype(np.float32)
    st_dev, _ = make_bn(rng, 64, dev)
    st = st_dev.cpu().numpy().astype(np.float64)
    dgamma, dbeta = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    sink = ops.BnSink(64, M, dev, dgamma=dgamma, dbeta=dbeta)
    out = torch.full((*outd, 64), float("nan"), device=dev)
    out2 = torch.full((*outd, 64), float("nan"), device=dev)
    kw = dict(out_mask=torch.from_numpy(act).to(dev), bwd=(torch.from_numpy(y).to(dev), st_dev, False), sink=sink,
              tail=(wdp, out2))
    assert ops.winograd_supported(g, **kw)
    ref = np.where(act > 0, conv_ref.conv_forward(dy, Wt, outd, k, s, p, mode=1), 0.0)
    ref2 = ref.reshape(M, 64) @ Wd[0].astype(np.float64)
    for rep in range(2):                                  # twice: the sink must come back to zero
        ops.conv_forward_winograd(g, torch.from_numpy(dy).to(dev), wu, out, **kw)
        torch.cuda.synchronize()
        assert rel_l2(out.cpu().numpy(), ref) <= TOL
        assert rel_l2(out2.cpu().numpy(), ref2) <= TOL
        yhat = (y.astype(np.float64).reshape(M, 64) - st[128:192]) * st[192:]
        db, dg = ref2.sum(0), (ref2 * yhat).sum(0)
        tol_s = 3e-6 * np.sqrt(M) * np.abs(ref2).max()
        assert np.abs(dbeta.cpu().numpy() - db).max() <= tol_s and np.abs(dgamma.cpu().numpy() - dg).max() <= 4 * tol_s, name


def test_lyft_network_direct_and_winograd_forms_agree(monkeypatch):
    """The whole Lyft-grid network (createModel, model_training.py:222-257) twice on the same variables and sweep: with the
    direct kernels everywhere (LISEC_TUNING winograd=0) and with the Winograd form where LisecNet runs it (winograd=7).  The
    training-mode forward maps and one training step's loss agree to fp32 rounding through 11 layers; the gradients of both
    against the fp64 oracle are the subject of tests/test_gpu_network.py, which runs the default (Winograd) form."""
    from conftest import LYFT
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-55, 55, 20000), rng.uniform(-55, 55, 20000), rng.uniform(-0.5, 2.5, 20000)], 1).astype(np.float32)
    op = M.glorot_params(seed=31, randomize_bn=True)
    y_cls = torch.from_numpy(rng.integers(0, 2, (100, 200, 2)).astype(np.float32)).to(dev)
    y_reg = torch.from_numpy(rng.normal(0, 1, (100, 200, 14)).astype(np.float32)).to(dev)
    res = {}
    for bits in (0, 7):
        monkeypatch.setenv("LISEC_TUNING", f"winograd={bits}")
        net = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
        assert bool(net.packed_wu) == (bits != 0)
        sample = Voxelizer(**LYFT)(pts)
        cls, reg = net.forward(sample, training=True)
        head = net.act["head"].cpu().numpy().copy()
        loss = net.train_step(sample, y_cls, y_reg)
        torch.cuda.synchronize()
        res[bits] = (head, float(loss[0].item()), net.params.grad_view(net.grad, "mid2.conv.kernel").cpu().numpy().copy())
    assert rel_l2(res[7][0], res[0][0]) <= 2e-5, "RPN maps of the two forms"
    assert abs(res[7][1] - res[0][1]) <= 1e-5 * abs(res[0][1]), "loss of the two forms"
    assert np.isfinite(res[7][2]).all() and np.abs(res[7][2]).max() > 0


TOL_W = 3e-6          # weight gradients sum 160 000 - 320 000 positions (the bound of tests/test_gpu_lyft_layers.py)


@pytest.mark.parametrize("case", [("small valid depth", (4, 12, 20), (2, 12, 20), 3, 1, 0),
                                  ("small odd map, depth stride 2 pad 1", (2, 11, 19), (1, 11, 19), 3, 2, 1),
                                  ("small 2d", (1, 34, 50), (1, 34, 50), 1, 1, 0),
                                  ("mid2", (4, H, W), (2, H, W), 3, 1, 0), ("mid3", (2, H, W), (1, H, W), 3, 2, 1)],
                         ids=["small", "small-odd", "small-2d", "mid2", "mid3"])
def test_weight_gradient_against_the_oracle(case):
    """dW[tap][c][n] = sum over positions of x[src(pos, tap)][c] dy[pos][n] in the Winograd form (lisec_conv_wgrad_winograd)
    against the fp64 restatement oracle/conv_ref.conv_wgrad -- the oracle and bound of the direct weight gradients."""
    from lisec_amd import ops
    from oracle import conv_ref
    name, ind, outd, KD, sd, pd = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(zlib.crc32(("wgrad" + name).encode()))
    k, s, p = (KD, 3, 3), (sd, 1, 1), (pd, 1, 1)
    x = rng.normal(0, 1, (*ind, 64)).astype(np.float32)
    dy = rng.normal(0, 1, (*outd, 64)).astype(np.float32)
    g = ops.geom(0, ind, outd, k, s, p, 64, 64)
    assert ops.wgrad_winograd_supported(g)
    ws = torch.empty(ops.wgrad_winograd_workspace_bytes(g), dtype=torch.uint8, device=dev)
    dW = torch.full((KD * 9, 64, 64), float("nan"), device=dev)
    ref = conv_ref.conv_wgrad(x, dy, outd, k, s, p, mode=0)
    for rep in range(2):
        ops.conv_wgrad_winograd(g, torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev), dW, ws)
        torch.cuda.synchronize()
        got = dW.cpu().numpy()
        assert np.isfinite(got).all()
        e = rel_l2(got, ref)
        assert e <= TOL_W, f"{name}: relative L2 {e:.2e}"
