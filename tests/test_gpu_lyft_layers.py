"""Every dense contraction of createModel at its REAL Lyft geometry (Constants.py: 8 x 200 x 400 grid), one layer at a time:
forward, data gradient and weight gradient through the C ABI against the fp64 restatement oracle/conv_ref.py (pinned
against torch's conv3d / conv2d / conv_transpose2d in tests/test_oracle_conv.py) on random, well-conditioned inputs.

Why: the op tests in test_gpu_conv.py / test_gpu_backward_ops.py use small maps, where the launch plans differ from the
ones the Lyft grid takes (plane pairing, 32-column workgroups, K slices combined in-kernel, parity classes, the resident
Dense(64) kernel, the tile queue, three-workgroup weight-gradient tiles); at the real geometry those plans were only
checked through whole-network tests whose gradient bound is loosened by BatchNormalization conditioning.  Here nothing
sits between input and output, so the bound is fp32 summation noise: relative L2 <= TOL (2e-6 forward / data gradient,
3e-6 for weight gradients, which sum 20 000 - 320 000 rows), and the plan every case engages is ASSERTED through
lisec_conv_plan_query / lisec_conv_wgrad_plan_query, not assumed.
Reference statements: model_training.py:193 (Conv3D), :195 (Dense 64), :203 (Conv2D), :246-252 (Conv2DTranspose),
:254-255 (heads), :299 (fit: the gradients)."""
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL, TOL_W = 2e-6, 3e-6
H, W = 200, 400


def rel_l2(got, ref, note=None):
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    e = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))
    if note and os.path.isdir("gpurun_out"):           # the measured errors, kept next to the profiles (scratch directory)
        with open(os.path.join("gpurun_out", "lyft_layer_errors.txt"), "a") as f:
            f.write(f"{note}: relative L2 {e:.3e}\n")
    return e


def seed_of(text):
    return zlib.crc32(text.encode())


def make_bn(rng, c, dev):
    """bnstate float[4C] {scale, shift, mean, invstd} of a producing layer"""
    scale, shift = rng.uniform(0.5, 1.5, c), rng.normal(0, 0.3, c)
    mean, invstd = rng.normal(0, 0.2, c), rng.uniform(0.7, 1.4, c)
    st = np.concatenate([scale, shift, mean, invstd]).astype(np.float32)
    return torch.from_numpy(st).to(dev), (st[:c].astype(np.float64), st[c:2 * c].astype(np.float64))


def check_plan(plan, expect):
    for k, v in expect.items():
        assert plan[k] == v, f"plan[{k}] = {plan[k]}, expected {v} ({plan})"


# name, mode, in dims, out dims, kernel, stride, pad, cin, cout, BN+ReLU on load, expected forward plan
FORWARD = [
    ("mid1 dense form", 0, (8, H, W), (4, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, False,
     dict(kernel="halo2", cols=64, k_slices=1, plane_pair=1, workgroups=1250)),
    ("mid2", 0, (4, H, W), (2, H, W), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, False,
     dict(kernel="halo2", cols=64, k_slices=1, plane_pair=0, workgroups=1250)),
    ("mid3", 0, (2, H, W), (1, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, False,
     dict(kernel="halo2", cols=64, k_slices=1, workgroups=625)),
    ("rpn1.conv0", 0, (1, H, W), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128, False,
     dict(kernel="igemm", cols=32, k_slices=1, workgroups=628)),
    ("rpn1.conv1", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True,
     dict(kernel="halo2", cols=32, k_slices=1, workgroups=628)),
    ("rpn1.conv1 [wide_tile=1]", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True,
     dict(kernel="wide", cols=128, k_slices=3, workgroups=471)),          # 128 x 128 tiles, 512-thread workgroups (opt-in)
    ("rpn2.conv0", 0, (1, 100, 200), (1, 50, 100), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128, True,
     dict(kernel="igemm", cols=64, k_slices=3, tail_tile0=0, workgroups=240, double_buffered=1)),
    ("rpn2.conv1", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True,
     dict(kernel="halo3", cols=64, k_slices=3, tail_tile0=0, workgroups=240, double_buffered=1)),
    ("rpn3.conv0", 0, (1, 50, 100), (1, 25, 50), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 256, True,
     dict(kernel="igemm", cols=64, k_slices=6, tail_tile0=0, workgroups=240, double_buffered=1)),
    ("rpn3.conv1", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, True,
     dict(kernel="igemm", cols=64, k_slices=6, tail_tile0=0, workgroups=240, double_buffered=1)),
]


@pytest.mark.parametrize("case", FORWARD, ids=[c[0] for c in FORWARD])
def test_forward_with_batch_statistics(case):
    """out = conv(f(x)) + bias as the training forward runs it: BatchNormalization + ReLU of the producer on load, the batch
    statistics of the result summed into a sink and finalised inside the call (bnstate)."""
    from lisec_amd import ops
    from oracle import conv_ref
    name, mode, ind, outd, k, s, p, cin, cout, xf, expect = case
    from lisec_amd import _lib
    _lib.set_tuning(wide_tile=1 if "[wide_tile=1]" in name else 0)
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed_of(name))
    ntaps = k[0] * k[1] * k[2]
    x = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
    Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    bn_dev, bn_ref = make_bn(rng, cin, dev) if xf else (None, None)
    M = outd[0] * outd[1] * outd[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(Wt).to(dev)
    wp = ops.pack_weights(wd, ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*outd, cout, device=dev)
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev)
    beta = torch.from_numpy(rng.normal(0, 0.2, cout).astype(np.float32)).to(dev)
    bnstate = torch.zeros(4 * cout, device=dev)
    sink = ops.BnSink(cout, M, dev, gamma=gamma, beta=beta, bnstate=bnstate)
    flags = ops.IN_RELU if xf else 0
    check_plan(ops.conv_plan(g, in_bn=xf, flags=flags, sink=sink), expect)
    ops.conv_forward(g, xd, wp, out, bias=torch.from_numpy(b).to(dev), in_bn=bn_dev, flags=flags, sink=sink)
    torch.cuda.synchronize()
    ref = conv_ref.conv_forward(x, Wt, outd, k, s, p, mode=mode, bias=b, in_bn=bn_ref, relu=xf)
    e = rel_l2(out.cpu().numpy(), ref, note="forward " + name)
    assert e <= TOL, f"{name}: forward relative L2 {e:.2e}"
    # batch statistics (Keras: biased variance, eps 1e-3) of the stored map
    y = ref.reshape(M, cout)
    mean, var = y.mean(0), y.var(0)
    inv = 1.0 / np.sqrt(var + 1e-3)
    st = bnstate.cpu().numpy().astype(np.float64)
    gm, bt = gamma.cpu().numpy().astype(np.float64), beta.cpu().numpy().astype(np.float64)
    assert np.allclose(st[2 * cout:3 * cout], mean, rtol=0, atol=2e-6 * np.abs(y).max()), f"{name}: batch mean"
    assert np.allclose(st[3 * cout:], inv, rtol=2e-5), f"{name}: inverse std"
    assert np.allclose(st[:cout], gm * inv, rtol=2e-5) and np.allclose(st[cout:2 * cout], bt - mean * gm * inv, rtol=1e-4,
                                                                         atol=1e-5), f"{name}: scale / shift"


# the data gradients the backward pass runs: dA of the PRODUCING layer = transposed contraction of dy with W^T.
# name, mode (of the data-gradient call), in dims (dy), out dims (dA), kernel, stride, pad, cin (= forward Cout), cout, plan
DGRAD = [
    ("mid2", 1, (2, H, W), (4, H, W), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, "mask",
     dict(kernel="halo2", cols=64, k_slices=1, plane_pair=1, workgroups=1250)),
    ("mid3", 1, (1, H, W), (2, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, "mask",
     dict(kernel="halo2", cols=64, k_slices=1, workgroups=1250)),
    ("rpn1.conv0", 1, (1, 100, 200), (1, H, W), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 64, "mask",
     dict(kernel="igemm", parity_classes=1)),
    ("rpn1.conv1", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, "bn",
     dict(kernel="halo2", cols=32, k_slices=1, workgroups=628)),
    ("rpn1.conv1 [wide_tile=1]", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, "bn",
     dict(kernel="wide", cols=128, k_slices=3, workgroups=471)),
    ("rpn2.conv0", 1, (1, 50, 100), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128, "bn",
     dict(kernel="igemm", parity_classes=1)),
    ("rpn2.conv1", 1, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, "bn",
     dict(kernel="halo3", cols=64, k_slices=3, workgroups=240, double_buffered=1)),
    ("rpn3.conv0", 1, (1, 25, 50), (1, 50, 100), (1, 3, 3), (1, 2, 2), (0, 1, 1), 256, 128, "bn",
     dict(kernel="igemm", parity_classes=1)),
    ("rpn3.conv1", 1, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, "bn",
     dict(kernel="igemm", cols=64, k_slices=6, workgroups=240, double_buffered=1)),
]


@pytest.mark.parametrize("case", DGRAD, ids=[c[0] for c in DGRAD])
def test_data_gradient(case):
    """dA = dy (*) W^T as dgrad_into() issues it.  'mask': the consumer-side ReLU of a middle block's Dense gates the store
    (out_mask); 'bn': the gradient is about to cross a BatchNormalization + ReLU backwards, and its (sum dz, sum dz*yhat)
    are summed into a backward sink and finalised inside the call (dbeta, dgamma, coefficients)."""
    from lisec_amd import ops
    from oracle import conv_ref
    name, mode, ind, outd, k, s, p, cin, cout, kind, expect = case
    from lisec_amd import _lib
    _lib.set_tuning(wide_tile=1 if "[wide_tile=1]" in name else 0)
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed_of("d" + name))
    ntaps = k[0] * k[1] * k[2]
    dy = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
    Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)      # (tap, forward Cout, forward Cin)
    M = outd[0] * outd[1] * outd[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.full((*outd, cout), float("nan"), device=dev)
    ref = conv_ref.conv_forward(dy, Wt, outd, k, s, p, mode=mode)
    kw = {}
    if kind == "mask":
        act = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
        kw["out_mask"] = torch.from_numpy(act).to(dev)
        ref = np.where(act > 0, ref, 0.0)
    else:
        y = rng.normal(0, 1, (*outd, cout)).astype(np.float32)                                  # raw output of the layer below
        st_dev, (scale, shift) = make_bn(rng, cout, dev)
        st = st_dev.cpu().numpy().astype(np.float64)
        dgamma, dbeta = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
        sink = ops.BnSink(cout, M, dev, dgamma=dgamma, dbeta=dbeta)
        kw.update(bwd=(torch.from_numpy(y).to(dev), st_dev, True), sink=sink)
    check_plan(ops.conv_plan(g, **{a: b for a, b in kw.items()}), expect)
    ops.conv_forward(g, torch.from_numpy(dy).to(dev), wp, out, **kw)
    torch.cuda.synchronize()
    e = rel_l2(out.cpu().numpy(), ref, note="data gradient " + name)
    assert e <= TOL, f"{name}: data gradient relative L2 {e:.2e}"
    if kind == "bn":
        y64 = y.astype(np.float64).reshape(M, cout)
        dz = np.where(y64 * scale + shift > 0, ref.reshape(M, cout), 0.0)
        yhat = (y64 - st[2 * cout:3 * cout]) * st[3 * cout:]
        db, dg = dz.sum(0), (dz * yhat).sum(0)
        tol_s = 3e-6 * np.sqrt(M) * np.abs(dz).max()                   # a sum of M terms of either sign
        assert np.abs(dbeta.cpu().numpy() - db).max() <= tol_s and np.abs(dgamma.cpu().numpy() - dg).max() <= 3 * tol_s, name
        coef = sink.coef.cpu().numpy().astype(np.float64)
        assert np.abs(coef[:cout] - db / M).max() <= tol_s / M and np.abs(coef[cout:] - dg / M).max() <= 3 * tol_s / M, name


@pytest.mark.parametrize("case", [c for c in DGRAD if c[0].startswith("rpn") and "[" not in c[0]],
                         ids=[c[0] for c in DGRAD if c[0].startswith("rpn") and "[" not in c[0]])
def test_data_gradient_with_bn_backward_on_load(case):
    """lisec_conv_extras.in_y: the gradient the contraction gathers is w.r.t. the OUTPUT of the layer's BatchNormalization +
    ReLU (model_training.py:204-206); the apply pass of that backward -- scale (gate(y) g - mean(dz) - yhat mean(dz yhat)) --
    runs on load, with the constants a backward sink finalised.  Against apply-then-contract in fp64, every RPN geometry."""
    from lisec_amd import _lib, ops
    from oracle import conv_ref
    name, mode, ind, outd, k, s, p, cin, cout, kind, expect = case
    _lib.set_tuning(wide_tile=0)
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed_of("fold" + name))
    ntaps = k[0] * k[1] * k[2]
    gq = rng.normal(0, 1, (*ind, cin)).astype(np.float32)              # gradient w.r.t. relu(bn(y))
    y = rng.normal(0, 1, (*ind, cin)).astype(np.float32)               # raw output of the layer the gradient belongs to
    st_dev, (scale, shift) = make_bn(rng, cin, dev)
    st = st_dev.cpu().numpy().astype(np.float64)
    mean, invstd = st[2 * cin:3 * cin], st[3 * cin:]
    y64, g64 = y.astype(np.float64), gq.astype(np.float64)
    dz = np.where(y64 * scale + shift > 0, g64, 0.0)
    yhat = (y64 - mean) * invstd
    m1, m2 = dz.reshape(-1, cin).mean(0), (dz * yhat).reshape(-1, cin).mean(0)
    coef = torch.from_numpy(np.concatenate([m1, m2]).astype(np.float32)).to(dev)
    c32 = coef.cpu().numpy().astype(np.float64)
    dy = scale * (dz - c32[:cin] - yhat * c32[cin:])                   # what lisec_bn_backward_apply_coef writes
    Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)
    geo = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), ntaps, cin, cout, cin * cout, cout, 1)
    out = torch.full((*outd, cout), float("nan"), device=dev)
    fold = (torch.from_numpy(y).to(dev), st_dev, coef, True)
    plan = ops.conv_plan(geo, fold=fold)
    check_plan(plan, {k_: v for k_, v in expect.items() if k_ in ("kernel", "cols", "k_slices", "parity_classes", "double_buffered")})
    ops.conv_forward(geo, torch.from_numpy(gq).to(dev), wp, out, fold=fold)
    torch.cuda.synchronize()
    ref = conv_ref.conv_forward(dy, Wt, outd, k, s, p, mode=mode)
    e = rel_l2(out.cpu().numpy(), ref, note="data gradient, BatchNormalization backward on load " + name)
    assert e <= TOL, f"{name}: relative L2 {e:.2e}"


@pytest.mark.parametrize("case", [c for c in DGRAD if c[0] in ("mid2", "mid3")], ids=["mid2", "mid3"])
def test_data_gradient_with_dense_tail(case):
    """The Dense(64, relu) data gradient of the block below riding on the tile (lisec_conv_extras.tail_w,
    model_training.py:195 backwards): out = gate(dy (*) W^T), out2 = out @ Wd^T, and (sum dz, sum dz*yhat) of out2 for the
    BatchNormalization under the Dense -- against the two separate steps in fp64."""
    from lisec_amd import ops
    from oracle import conv_ref
    name, mode, ind, outd, k, s, p, cin, cout, kind, expect = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed_of("tail" + name))
    ntaps = k[0] * k[1] * k[2]
    dy = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
    Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)
    Wd = (rng.normal(0, 1, (1, 64, 64)) / 8).astype(np.float32)          # the tail kernel as the contraction sees it: (c, j)
    M = outd[0] * outd[1] * outd[2]
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), ntaps, cin, cout, cin * cout, cout, 1)
    wdp = ops.pack_weights(torch.from_numpy(Wd).to(dev), 1, 64, 64, 0, 64, 1)
    act = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
    y = rng.normal(0, 1, (*outd, 64)).astype(np.float32)
    st_dev, _ = make_bn(rng, 64, dev)
    st = st_dev.cpu().numpy().astype(np.float64)
    dgamma, dbeta = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    sink = ops.BnSink(64, M, dev, dgamma=dgamma, dbeta=dbeta)
    out = torch.full((*outd, cout), float("nan"), device=dev)
    out2 = torch.full((*outd, 64), float("nan"), device=dev)
    kw = dict(out_mask=torch.from_numpy(act).to(dev), bwd=(torch.from_numpy(y).to(dev), st_dev, False), sink=sink,
              tail=(wdp, out2))
    check_plan(ops.conv_plan(g, **kw), dict(kernel="halo2", cols=64, k_slices=1, workgroups=1250))
    for rep in range(2):                                  # twice: the sink must come back to zero
        ops.conv_forward(g, torch.from_numpy(dy).to(dev), wp, out, **kw)
        torch.cuda.synchronize()
        ref = np.where(act > 0, conv_ref.conv_forward(dy, Wt, outd, k, s, p, mode=mode), 0.0)
        assert rel_l2(out.cpu().numpy(), ref, note="data gradient with tail " + name) <= TOL
        ref2 = ref.reshape(M, 64) @ Wd[0].astype(np.float64)
        assert rel_l2(out2.cpu().numpy(), ref2, note="dense tail " + name) <= TOL
        yhat = (y.astype(np.float64).reshape(M, 64) - st[128:192]) * st[192:]
        db, dg = ref2.sum(0), (ref2 * yhat).sum(0)
        tol_s = 3e-6 * np.sqrt(M) * np.abs(ref2).max()
        assert np.abs(dbeta.cpu().numpy() - db).max() <= tol_s and np.abs(dgamma.cpu().numpy() - dg).max() <= 4 * tol_s, name
        coef = sink.coef.cpu().numpy().astype(np.float64)
        assert np.abs(coef[:64] - db / M).max() <= tol_s / M and np.abs(coef[64:] - dg / M).max() <= 4 * tol_s / M, name


# weight gradients: name, mode, in dims, out dims, kernel, stride, pad, cin, cout, BN+ReLU on load, transpose_out, plan
WGRAD = [
    ("mid1 dense form", 0, (8, H, W), (4, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, False, False,
     dict(ring=1, halo=0, groups=11, tile_rows=100, staging_passes=5, taps_per_group=9, lane_reduce=1)),   # kd = 0 reads no plane at d = 0
    ("mid2", 0, (4, H, W), (2, H, W), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, False, False,
     dict(ring=1, halo=0, groups=6, tile_rows=100, staging_passes=5, taps_per_group=9, lane_reduce=1, runs_per_column=10,
          workgroups=240)),
    ("mid3", 0, (2, H, W), (1, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, False, False,
     dict(ring=1, groups=2, tile_rows=100, staging_passes=5, lane_reduce=1)),      # the kd = 0 cells read nothing: zeros
    ("mid1.dense", 0, (4, H, W), (4, H, W), (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64, True, False,
     dict(ring=0, halo=0, taps_per_group=1, lane_reduce=1)),
    ("rpn1.conv0", 0, (1, H, W), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128, False, False,
     dict(ring=0, halo=0, taps_per_group=3, groups=3)),
    ("rpn1.conv1", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True, False,
     dict(ring=1, groups=1, tile_rows=100, staging_passes=5, runs_per_column=25, lane_reduce=1)),   # alone: 50 slices per cell
    ("rpn2.conv0", 0, (1, 100, 200), (1, 50, 100), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 128, True, False,
     dict(ring=0, halo=0, taps_per_group=3)),
    ("rpn2.conv1", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, True, False,
     dict(ring=1, groups=1, tile_rows=100, staging_passes=5, lane_reduce=1)),
    ("rpn3.conv0", 0, (1, 50, 100), (1, 25, 50), (1, 3, 3), (1, 2, 2), (0, 1, 1), 128, 256, True, False,
     dict(ring=0, halo=0, taps_per_group=3)),
    ("rpn3.conv1", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, True, False,
     dict(ring=1, groups=1, tile_rows=50, staging_passes=3, combine_in_kernel=1)),
    ("up1 (256 channels)", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256, True, True,
     dict(ring=1, mirrored=1, groups=1, tile_rows=100)),
    ("up1 collapsed (16 channels)", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 16, True, False,
     dict(ring=1, mirrored=1, groups=1, tile_rows=100)),
    ("up2 collapsed (1x1, 64 columns)", 0, (1, 50, 100), (1, 50, 100), (1, 1, 1), (1, 1, 1), (0, 0, 0), 128, 64, True, False,
     dict(ring=0, halo=0, taps_per_group=1)),
    ("up3 collapsed (1x1, 256 columns)", 0, (1, 25, 50), (1, 25, 50), (1, 1, 1), (1, 1, 1), (0, 0, 0), 256, 256, True, False,
     dict(ring=0, halo=0, taps_per_group=1)),
]


@pytest.mark.parametrize("case", WGRAD, ids=[c[0] for c in WGRAD])
def test_weight_gradient(case):
    from lisec_amd import ops
    from oracle import conv_ref
    name, mode, ind, outd, k, s, p, cin, cout, xf, transpose_out, expect = case
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed_of("w" + name))
    ntaps = k[0] * k[1] * k[2]
    x = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
    dy = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
    bn_dev, bn_ref = make_bn(rng, cin, dev) if xf else (None, None)
    g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
    flags = ops.IN_RELU if xf else 0
    check_plan(ops.wgrad_plan(g, flags=flags), expect)
    ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=dev)
    dW = torch.full((ntaps, cout, cin) if transpose_out else (ntaps, cin, cout), float("nan"), device=dev)
    ops.conv_wgrad(g, torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev), dW, ws, in_bn=bn_dev, flags=flags,
                   transpose_out=transpose_out)
    torch.cuda.synchronize()
    ref = conv_ref.conv_wgrad(x, dy, outd, k, s, p, mode=mode, in_bn=bn_ref, relu=xf)
    got = dW.cpu().numpy()
    if transpose_out:
        got = np.transpose(got, (0, 2, 1))
    e = rel_l2(got, ref, note="weight gradient " + name)
    assert e <= TOL_W, f"{name}: weight gradient relative L2 {e:.2e}"


def test_dense64_resident_kernel_forward_and_data_gradient():
    """Dense(64, relu) on the BatchNormalization output of a middle block (model_training.py:194-195) at 320 000 positions:
    the resident-workgroup kernel, forward (BN on load, ReLU on store) and data gradient with backward statistics."""
    from lisec_amd import ops
    from oracle import conv_ref
    dev = torch.device("cuda")
    rng = np.random.default_rng(64)
    dims, M = (4, H, W), 4 * H * W
    x = rng.normal(0, 1, (*dims, 64)).astype(np.float32)
    Wt = (rng.normal(0, 1, (1, 64, 64)) / 8).astype(np.float32)
    bn_dev, bn_ref = make_bn(rng, 64, dev)
    g = ops.geom(0, dims, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64)
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), 1, 64, 64, 0, 64, 1)
    out = torch.empty(*dims, 64, device=dev)
    check_plan(ops.conv_plan(g, in_bn=True, flags=ops.OUT_RELU), dict(kernel="dense64", workgroups=768))
    ops.conv_forward(g, torch.from_numpy(x).to(dev), wp, out, in_bn=bn_dev, flags=ops.OUT_RELU)
    ref = np.maximum(conv_ref.conv_forward(x, Wt, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0), in_bn=bn_ref), 0.0)
    assert rel_l2(out.cpu().numpy(), ref) <= TOL
    # data gradient: dz = du @ Wd^T, (sum dz, sum dz*yhat) of the BatchNormalization under it into a backward sink
    du = rng.normal(0, 1, (*dims, 64)).astype(np.float32)
    wpt = ops.pack_weights(torch.from_numpy(Wt).to(dev), 1, 64, 64, 0, 1, 64)
    st = bn_dev.cpu().numpy().astype(np.float64)
    dgamma, dbeta = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    sink = ops.BnSink(64, M, dev, dgamma=dgamma, dbeta=dbeta)
    dz = torch.empty(*dims, 64, device=dev)
    xd = torch.from_numpy(x).to(dev)
    check_plan(ops.conv_plan(g, bwd=(xd, bn_dev, False), sink=sink), dict(kernel="dense64"))
    ops.conv_forward(g, torch.from_numpy(du).to(dev), wpt, dz, bwd=(xd, bn_dev, False), sink=sink)
    torch.cuda.synchronize()
    ref_dz = conv_ref.conv_forward(du, np.transpose(Wt, (0, 2, 1)), dims, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    assert rel_l2(dz.cpu().numpy(), ref_dz) <= TOL
    yhat = (x.astype(np.float64).reshape(M, 64) - st[128:192]) * st[192:]
    r = ref_dz.reshape(M, 64)
    tol_s = 3e-6 * np.sqrt(M) * np.abs(r).max()
    assert np.abs(dbeta.cpu().numpy() - r.sum(0)).max() <= tol_s
    assert np.abs(dgamma.cpu().numpy() - (r * yhat).sum(0)).max() <= 4 * tol_s


@pytest.mark.parametrize("dims,relu", [((1, H, W), False), ((2, H, W), False), ((1, H - 1, W + 1), True)])
def test_dense64_data_gradient_carries_the_weight_gradient(dims, relu):
    """Dense(64) backwards in ONE pass over the two maps (model_training.py:195; what fit() derives for the Dense kernel, :299):
    the data-gradient call reads the gradient rows (its A operand) and the rows of y (its backward statistics), which are the
    two operands of dW = bn(y)^T du -- lisec_conv_extras.dense_dw.  Against the fp64 oracle, against lisec_conv_wgrad, bit for
    bit against itself; the ragged case ends in a partial tile."""
    from lisec_amd import ops, _lib
    from oracle import conv_ref
    dev = torch.device("cuda")
    rng = np.random.default_rng(dims[0] * 7 + dims[1])
    M = dims[0] * dims[1] * dims[2]
    y = rng.normal(0, 1, (*dims, 64)).astype(np.float32)
    du = rng.normal(0, 1, (*dims, 64)).astype(np.float32)
    Wt = (rng.normal(0, 1, (1, 64, 64)) / 8).astype(np.float32)
    bn_dev, bn_ref = make_bn(rng, 64, dev)
    g = ops.geom(0, dims, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64)
    wpt = ops.pack_weights(torch.from_numpy(Wt).to(dev), 1, 64, 64, 0, 1, 64)
    yd, dud = torch.from_numpy(y).to(dev), torch.from_numpy(du).to(dev)
    n = ops.dense_dw_slabs()
    assert n == 512
    results = []
    for rep in range(2):
        dgamma, dbeta = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        sink = ops.BnSink(64, M, dev, dgamma=dgamma, dbeta=dbeta)
        slabs = torch.full((n * 4096,), float("nan"), device=dev)
        dz = torch.empty(*dims, 64, device=dev)
        dW = torch.full((64, 64), float("nan"), device=dev)
        check_plan(ops.conv_plan(g, bwd=(yd, bn_dev, relu), sink=sink, dense_dw=slabs), dict(kernel="dense64", workgroups=n))
        ops.conv_forward(g, dud, wpt, dz, bwd=(yd, bn_dev, relu), sink=sink, dense_dw=slabs)
        ops.dense_dw_reduce(slabs, dW)
        torch.cuda.synchronize()
        results.append((dz.cpu().numpy(), dW.cpu().numpy(), dgamma.cpu().numpy(), dbeta.cpu().numpy()))
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)
    dz_h, dW_h, dgamma_h, dbeta_h = results[0]
    ref_dz = conv_ref.conv_forward(du, np.transpose(Wt, (0, 2, 1)), dims, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    assert rel_l2(dz_h, ref_dz) <= TOL
    ref_dW = conv_ref.conv_wgrad(y, du, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0), mode=0, in_bn=bn_ref, relu=relu)
    e = rel_l2(dW_h[None], ref_dW, note="Dense weight gradient beside the data gradient")
    assert e <= TOL_W, f"relative L2 {e:.2e}"
    ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=dev)
    dW2 = torch.empty(1, 64, 64, device=dev)
    ops.conv_wgrad(g, yd, dud, dW2, ws, in_bn=bn_dev, flags=ops.IN_RELU if relu else 0)
    assert rel_l2(dW_h[None], dW2.cpu().numpy()) <= TOL_W
    st = bn_dev.cpu().numpy().astype(np.float64)
    yhat = (y.astype(np.float64).reshape(M, 64) - st[128:192]) * st[192:]
    r = ref_dz.reshape(M, 64)
    if relu:
        r = r * ((y.astype(np.float64).reshape(M, 64) * st[:64] + st[64:128]) > 0)
    tol_s = 3e-6 * np.sqrt(M) * np.abs(r).max()
    assert np.abs(dbeta_h - r.sum(0)).max() <= tol_s
    assert np.abs(dgamma_h - (r * yhat).sum(0)).max() <= 4 * tol_s
    # refused where the call is not that data gradient: too few tiles, no backward statistics
    small = ops.geom(0, (1, 100, 200), (1, 100, 200), (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64)
    with pytest.raises(_lib.LisecError):
        ops.conv_forward(small, dud, wpt, dz, bwd=(yd, bn_dev, relu), sink=sink, dense_dw=slabs)
    with pytest.raises(_lib.LisecError):
        ops.conv_forward(g, dud, wpt, dz, dense_dw=slabs)


@pytest.mark.parametrize("cap,expect_kernel", [(20000, "igemm"), (120000, "queue")])
def test_first_conv3d_row_list_gradients(cap, expect_kernel):
    """Exact sparse backward of the first Conv3D (model_training.py:236): its data gradient evaluated only at the occupied
    cells (a row list of voxel coordinates; large capacities draw their tiles from a queue) and the voxel part of its weight
    gradient, sum_v delta_v (x) dy[q(p_v, tap)]."""
    from lisec_amd import ops
    from oracle import conv_ref
    dev = torch.device("cuda")
    rng = np.random.default_rng(cap)
    V = cap - 137                                        # fewer rows than the capacity: the tail tiles hold nothing
    cells = rng.choice(8 * H * W, V, replace=False)
    cells.sort()
    coords = np.stack([cells // (H * W), (cells // W) % H, cells % W], 1).astype(np.int32)
    dz = rng.normal(0, 1, (4, H, W, 64)).astype(np.float32)
    Wt = (rng.normal(0, 1, (27, 64, 64)) / np.sqrt(27 * 64)).astype(np.float32)                 # (tap, forward Cout, forward Cin)
    dg = ops.geom(1, (4, H, W), (8, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64)
    wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), 27, 64, 64, 64 * 64, 64, 1)
    coords_d = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
    coords_d[:V] = torch.from_numpy(coords).to(dev)
    count = torch.tensor([V, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    out = torch.full((cap + 1, 64), float("nan"), device=dev)
    queue = torch.zeros(2, dtype=torch.int32, device=dev)
    check_plan(ops.conv_plan(dg, rows_capacity=cap, queue=queue), dict(kernel=expect_kernel))
    dzd = torch.from_numpy(dz).to(dev)
    ops.conv_forward(dg, dzd, wp, out, rows=(coords_d, count, cap), queue=queue)
    torch.cuda.synchronize()
    ref = conv_ref.conv_forward(dz, Wt, (8, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), mode=1, rows=coords)
    assert rel_l2(out[:V].cpu().numpy(), ref) <= TOL
    assert int(queue.abs().sum().item()) == 0            # the tile counter is left at zero for the next call
    # weight gradient over the row list: dW[tap][c][n] = sum_v dz[src(p_v, tap)][c] * delta[v][n], stored (tap, n, c)
    delta = rng.normal(0, 1, (cap + 1, 64)).astype(np.float32)
    ws = torch.zeros(ops.wgrad_workspace_bytes(dg, cap), dtype=torch.uint8, device=dev)
    dW = torch.full((27, 64, 64), float("nan"), device=dev)
    ops.conv_wgrad(dg, dzd, torch.from_numpy(delta).to(dev), dW, ws, transpose_out=True, rows=(coords_d, count, cap))
    torch.cuda.synchronize()
    ref_w = conv_ref.conv_wgrad(dz, delta[:V], (8, H, W), (3, 3, 3), (2, 1, 1), (1, 1, 1), mode=1, rows=coords)
    assert rel_l2(np.transpose(dW.cpu().numpy(), (0, 2, 1)), ref_w) <= TOL_W


def test_collapsed_branch_contractions():
    """The 16-channel contractions of the collapsed upsampling branches + heads (csrc/head_fused.hip) at the Lyft maps:
    forward of the 3x3 branch straight into the (100,200,16) head with the composite bias, the two kernel == stride branches
    as 1x1 contractions with (tap, j) columns, and the data gradients back to the 128- / 256-channel branch inputs."""
    from lisec_amd import ops
    from oracle import conv_ref
    dev = torch.device("cuda")
    rng = np.random.default_rng(16)
    for name, mode, ind, outd, k, s, p, cin, cout in [
            ("up1'", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 16),
            ("up2'", 0, (1, 50, 100), (1, 50, 100), (1, 1, 1), (1, 1, 1), (0, 0, 0), 128, 64),
            ("up3'", 0, (1, 25, 50), (1, 25, 50), (1, 1, 1), (1, 1, 1), (0, 0, 0), 256, 256)]:
        ntaps = k[1] * k[2]
        x = rng.normal(0, 1, (*ind, cin)).astype(np.float32)
        Wt = (rng.normal(0, 1, (ntaps, cin, cout)) / np.sqrt(ntaps * cin)).astype(np.float32)
        b = rng.normal(0, 0.1, cout).astype(np.float32)
        bn_dev, bn_ref = make_bn(rng, cin, dev)
        g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
        wp = ops.pack_weights(torch.from_numpy(Wt).to(dev), ntaps, cin, cout, cin * cout, cout, 1)
        out = torch.empty(*outd, cout, device=dev)
        ops.conv_forward(g, torch.from_numpy(x).to(dev), wp, out, bias=torch.from_numpy(b).to(dev), in_bn=bn_dev,
                         flags=ops.IN_RELU)
        ref = conv_ref.conv_forward(x, Wt, outd, k, s, p, mode=mode, bias=b, in_bn=bn_ref, relu=True)
        assert rel_l2(out.cpu().numpy(), ref) <= TOL, name
        # data gradient back to the branch input: mode flipped, kernel transposed
        dy = rng.normal(0, 1, (*outd, cout)).astype(np.float32)
        dgeo = ops.geom(1 - mode if ntaps > 1 else 0, outd, ind, k, s, p, cout, cin)
        wpt = ops.pack_weights(torch.from_numpy(Wt).to(dev), ntaps, cout, cin, cin * cout, 1, cout)
        dx = torch.empty(*ind, cin, device=dev)
        ops.conv_forward(dgeo, torch.from_numpy(dy).to(dev), wpt, dx)
        torch.cuda.synchronize()
        ref_dx = conv_ref.conv_forward(dy, np.transpose(Wt, (0, 2, 1)), ind, k, s, p, mode=1 - mode if ntaps > 1 else 0)
        assert rel_l2(dx.cpu().numpy(), ref_dx) <= TOL, name + " data gradient"


@pytest.mark.parametrize("block", ["rpn3", "rpn2", "rpn1"])
def test_batched_weight_gradients_of_an_rpn_block(block):
    """lisec_conv_wgrad_batched: the stride-1 convolutions of one RPN block in ONE launch (slabs summed in-kernel by the last
    slice of every cell) must give what the one-by-one launches give -- compared against the fp64 reference, each layer."""
    from lisec_amd import ops
    from oracle import conv_ref
    dev = torch.device("cuda")
    n, hw, ch = {"rpn3": (5, (25, 50), 256), "rpn2": (5, (50, 100), 128), "rpn1": (3, (100, 200), 128)}[block]
    rng = np.random.default_rng(seed_of("batch" + block))
    dims = (1, *hw)
    g = ops.geom(0, dims, dims, (1, 3, 3), (1, 1, 1), (0, 1, 1), ch, ch)
    items, host = [], []
    for _ in range(n):
        x = rng.normal(0, 1, (*dims, ch)).astype(np.float32)
        dy = rng.normal(0, 1, (*dims, ch)).astype(np.float32)
        bn_dev, bn_ref = make_bn(rng, ch, dev)
        dW = torch.full((9, ch, ch), float("nan"), device=dev)
        items.append((g, torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev), dW, bn_dev, ops.IN_RELU, False))
        host.append((x, dy, bn_ref))
    batch = ops.WgradBatch(items)
    ws = torch.zeros(batch.workspace_bytes(), dtype=torch.uint8, device=dev)
    for rep in range(2):                                   # twice: the arrival counters must come back to zero
        for it in items:
            it[3].fill_(float("nan"))
        batch.run(ws)
        torch.cuda.synchronize()
        for i, (x, dy, bn_ref) in enumerate(host):
            ref = conv_ref.conv_wgrad(x, dy, dims, (1, 3, 3), (1, 1, 1), (0, 1, 1), in_bn=bn_ref, relu=True)
            e = rel_l2(items[i][3].cpu().numpy(), ref, note=f"batched weight gradient {block} layer {i} run {rep}")
            assert e <= TOL_W, f"{block} layer {i} run {rep}: relative L2 {e:.2e}"
    head = ws[:4096 * 4].view(torch.int32)
    assert int(head.abs().sum().item()) == 0
