"""Backward building blocks (C ABI) vs torch autograd on CPU (fp32 reference, fp64 for reductions)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, rtol=1e-4):
    got, ref = got.double().cpu(), ref.double().cpu()
    atol = rtol * max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs()
    assert (err <= atol + rtol * ref.abs()).all(), f"max err {err.max().item():.3e}, atol {atol:.3e}"


def _wgrad_case(D, H, W, Cin, Cout, k, stride, pad, in_bn, in_relu, seed):
    from lisec_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(D, H, W, Cin, generator=g)
    w = (torch.randn(*k, Cin, Cout, generator=g) * 0.1).requires_grad_(True)
    xin, bn = x, None
    if in_bn:
        sc, sh = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)
        bn = torch.cat([sc, sh, torch.zeros(2 * Cin)]).to(DEV)
        xin = x * sc + sh
    if in_relu:
        xin = F.relu(xin)
    y = F.conv3d(xin.permute(3, 0, 1, 2)[None], w.permute(4, 3, 0, 1, 2), None, stride=stride, padding=pad)[0]
    y = y.permute(1, 2, 3, 0)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Do, Ho, Wo = y.shape[:3]
    geo = ops.geom(0, (D, H, W), (Do, Ho, Wo), k, stride, pad, Cin, Cout)
    ws = torch.zeros(ops.wgrad_workspace_bytes(geo), dtype=torch.uint8, device=DEV)   # zero-filled once (arrival counters)
    dW = torch.full(w.shape, float("nan"), device=DEV)
    ops.conv_wgrad(geo, x.to(DEV), dy.to(DEV), dW, ws, in_bn=bn, flags=ops.IN_RELU if in_relu else 0)
    _close(dW, w.grad)


def test_wgrad_conv3d_and_conv2d():
    _wgrad_case(8, 12, 20, 64, 64, (3, 3, 3), (2, 1, 1), (1, 1, 1), False, False, 0)
    _wgrad_case(4, 9, 21, 64, 64, (3, 3, 3), (1, 1, 1), (0, 1, 1), False, False, 1)
    _wgrad_case(1, 24, 40, 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), False, False, 2)
    _wgrad_case(1, 12, 20, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, True, 3)
    _wgrad_case(1, 5, 7, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, True, 4)
    _wgrad_case(2, 10, 30, 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, False, 5)      # Dense on BN output
    _wgrad_case(1, 10, 30, 768, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, False, 6)    # heads


@pytest.mark.parametrize("k,s,cin", [(3, 1, 128), (2, 2, 128), (4, 4, 256)])
def test_wgrad_and_dgrad_conv2d_transpose(k, s, cin):
    from lisec_amd import ops
    g = torch.Generator().manual_seed(k + 10)
    H, W, cout = 6, 10, 256
    x = torch.randn(H, W, cin, generator=g, requires_grad=True)
    w = (torch.randn(k, k, cout, cin, generator=g) * 0.1).requires_grad_(True)     # (kh,kw,out,in)
    sc, sh = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
    a = F.relu(x * sc + sh)
    a.retain_grad()
    pad = (k - s) // 2
    y = F.conv_transpose2d(a.permute(2, 0, 1)[None], w.permute(3, 2, 0, 1), None, stride=s, padding=pad)[0].permute(1, 2, 0)
    Ho, Wo = H * s, W * s
    dcat = torch.randn(Ho, Wo, 768, generator=g)
    y.backward(dcat[:, :, 512:])
    geo = ops.geom(1, (1, H, W), (1, Ho, Wo), (1, k, k), (1, s, s), (0, pad, pad), cin, cout, out_stride=768)
    ws = torch.zeros(ops.wgrad_workspace_bytes(geo), dtype=torch.uint8, device=DEV)   # zero-filled once (arrival counters)
    dW = torch.full(w.shape, float("nan"), device=DEV)
    bn = torch.cat([sc, sh, torch.zeros(2 * cin)]).to(DEV)
    dcat_d = dcat.to(DEV)
    ops.conv_wgrad(geo, x.detach().to(DEV), dcat_d[:, :, 512:], dW, ws, in_bn=bn, flags=ops.IN_RELU, transpose_out=True)
    _close(dW, w.grad)
    # data gradient wrt a: mode-0 conv over dY with stride s, K = out, N = in
    geo_d = ops.geom(0, (1, Ho, Wo), (1, H, W), (1, k, k), (1, s, s), (0, pad, pad), cout, cin, in_stride=768)
    if k == s:
        # kernel == stride: weight gradient with swapped roles (gather dY, contract against relu(bn(x)))
        ws2 = torch.zeros(ops.wgrad_workspace_bytes(geo_d), dtype=torch.uint8, device=DEV)
        dW2 = torch.full(w.shape, float("nan"), device=DEV)
        ops.conv_wgrad(geo_d, dcat_d[:, :, 512:], x.detach().to(DEV), dW2, ws2, flags=ops.DY_RELU, dy_bn=bn)
        _close(dW2, w.grad)
    wp = ops.pack_weights(w.detach().to(DEV), k * k, cout, cin, cout * cin, cin, 1)
    da = torch.full((H, W, cin), float("nan"), device=DEV)
    ops.conv_forward(geo_d, dcat_d[:, :, 512:], wp, da)
    _close(da, a.grad)


@pytest.mark.parametrize("C,relu", [(64, False), (128, True), (256, True)])
def test_bn_backward(C, relu):
    from lisec_amd import ops
    g = torch.Generator().manual_seed(C)
    M = 3001
    y = (torch.randn(M, C, generator=g) * 2 + 0.5).double().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).double().requires_grad_(True)
    beta = torch.randn(C, generator=g).double().requires_grad_(True)
    mean = y.mean(0)
    var = ((y - mean) ** 2).mean(0)
    inv = gamma * torch.rsqrt(var + 1e-3)
    z = y * inv + (beta - mean * inv)
    a = F.relu(z) if relu else z
    dA = torch.randn(M, C + 64, generator=g)                       # strided gradient source
    a.backward(dA[:, :C].double())
    st = torch.cat([inv, beta - mean * inv, mean, torch.rsqrt(var + 1e-3)]).float().detach().to(DEV)
    dg, db, dbias = (torch.empty(C, device=DEV) for _ in range(3))
    dy = torch.empty(M, C, device=DEV)
    ops.bn_backward(dA.to(DEV), C + 64, y.detach().float().to(DEV), st, M, C, relu, dg, db, dy, dbias=dbias)
    _close(dy, y.grad, rtol=2e-4)
    _close(dg, gamma.grad, rtol=2e-4)
    _close(db, beta.grad, rtol=2e-4)
    assert dbias.abs().max().item() < 1e-2                        # sum of dy is zero up to rounding


def test_relu_mask_colsum_loss_sgd():
    from lisec_amd import ops
    g = torch.Generator().manual_seed(3)
    u = torch.randn(1000, 64, generator=g)
    d = torch.randn(1000, 64, generator=g)
    dd = d.to(DEV)
    ops.relu_mask(dd, u.to(DEV))
    assert torch.equal(dd.cpu(), d * (u > 0))
    x = torch.randn(777, 768, generator=g)
    out = torch.empty(256, device=DEV)
    ops.colsum(x.to(DEV)[:, 256:], 768, 777, 256, out)
    _close(out, x[:, 256:512].double().sum(0), rtol=1e-5)
    out16 = torch.empty(16, device=DEV)
    ops.colsum(x.to(DEV), 768, 777, 16, out16)
    _close(out16, x[:, :16].double().sum(0), rtol=1e-5)
    # losses
    M = 20 * 40
    head = torch.randn(M, 16, generator=g).requires_grad_(True)
    yc = torch.randint(0, 3, (M, 2), generator=g).float()
    yr = torch.randn(M, 14, generator=g) * 2
    for kind in (0, 1):
        head.grad = None
        if kind == 0:
            lc, lr = ((head[:, :2] - yc) ** 2).mean(), ((head[:, 2:] - yr) ** 2).mean()
        else:
            lc = F.binary_cross_entropy_with_logits(head[:, :2], yc.clamp(0, 1))
            lr = F.smooth_l1_loss(head[:, 2:], yr)
        (lc + lr).backward()
        dh = torch.empty(M, 16, device=DEV)
        lo = torch.empty(3, device=DEV)
        ops.rpn_loss(head.detach().to(DEV), yc.to(DEV), yr.to(DEV), M, kind, dh, lo)
        _close(dh, head.grad, rtol=1e-5)
        _close(lo, torch.stack([lc + lr, lc, lr]).detach(), rtol=1e-5)
    # SGD nesterov, two steps with decay (Keras semantics)
    w = torch.randn(4096, generator=g)
    v = torch.zeros(4096)
    wd, vd = w.to(DEV), v.to(DEV)
    for it in range(2):
        gr = torch.randn(4096, generator=g)
        lr_t = 0.01 / (1 + 1e-6 * it)
        v = 0.9 * v - lr_t * gr
        w = w + 0.9 * v - lr_t * gr
        ops.sgd_nesterov_step(wd, gr.to(DEV), vd, lr_t, 0.9)
    _close(wd, w, rtol=1e-6)
    _close(vd, v, rtol=1e-6)


@pytest.mark.parametrize("M", [1250, 5000, 9001])
@pytest.mark.parametrize("C,relu", [(128, True), (256, True), (64, False)])
def test_bn_backward_in_place_without_bias(M, C, relu):
    """The RPN layers' call: no conv-bias gradient, dy written over dA (the sizes of RPN blocks 3 and 2, and a ragged one)."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(C + M)
    y = (torch.randn(M, C, generator=g) * 2 + 0.5).double().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).double().requires_grad_(True)
    beta = torch.randn(C, generator=g).double().requires_grad_(True)
    mean = y.mean(0)
    var = ((y - mean) ** 2).mean(0)
    inv = gamma * torch.rsqrt(var + 1e-3)
    z = y * inv + (beta - mean * inv)
    a = F.relu(z) if relu else z
    dA = torch.randn(M, C, generator=g)
    a.backward(dA.double())
    st = torch.cat([inv, beta - mean * inv, mean, torch.rsqrt(var + 1e-3)]).float().detach().to(DEV)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    buf = dA.to(DEV).clone()
    ops.bn_backward(buf, C, y.detach().float().to(DEV), st, M, C, relu, dg, db, buf)
    _close(buf, y.grad, rtol=2e-4)
    _close(dg, gamma.grad, rtol=2e-4)
    _close(db, beta.grad, rtol=2e-4)


@pytest.mark.parametrize("dims", [((8, 16, 24), (4, 16, 24), (2, 1, 1), (1, 1, 1)), ((4, 12, 40), (2, 12, 40), (1, 1, 1), (0, 1, 1))])
def test_tap_sums_alone_and_fused_with_the_bn_backward_apply(dims):
    """lisec_conv_tap_sums: S[tap][n] = sum of dy over the output positions whose tap reads inside the input map (a
    torch transposed convolution of dy against ones, summed over space, is the same number); lisec_conv_tap_sums_bn:
    the same sums over dy = bn_backward_apply_coef(dz) with dy written in place -- equal to the two separate calls."""
    from lisec_amd import ops
    ind, outd, stride, pad = dims
    g = torch.Generator().manual_seed(23)
    C = 64
    geo = ops.geom(0, ind, outd, (3, 3, 3), stride, pad, C, C)
    M = outd[0] * outd[1] * outd[2]
    dz = torch.randn(M, C, generator=g).to(DEV)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.2).to(DEV)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = torch.rsqrt(var + 1e-3)
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV)
    st = torch.cat([gamma * inv, beta - mean * gamma * inv, mean, inv]).contiguous()
    coef = torch.cat([dz.mean(0), (dz * (y - mean) * inv).mean(0)]).contiguous()
    ws = torch.empty(ops.tap_sums_workspace_bytes(geo), dtype=torch.uint8, device=DEV)
    want_dy = torch.empty_like(dz)
    ops.bn_backward_apply_coef(dz, C, y, st, M, C, False, coef, want_dy)
    S_sep = torch.empty(27, C, device=DEV)
    ops.tap_sums(geo, want_dy, S_sep, ws)
    # independent value: does tap (kd,kh,kw) of output position o read inside the map?
    dyv = want_dy.double().cpu().reshape(*outd, C)
    ref = torch.zeros(27, C, dtype=torch.float64)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                ok = [torch.tensor([0 <= o * s - p + k < n for o in range(no)])
                      for k, s, p, n, no in zip((kd, kh, kw), stride, pad, ind, outd)]
                m = ok[0][:, None, None] & ok[1][None, :, None] & ok[2][None, None, :]
                ref[(kd * 3 + kh) * 3 + kw] = dyv[m].sum(0)
    np.testing.assert_allclose(S_sep.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    got = dz.clone()
    S_fused = torch.empty(27, C, device=DEV)
    ops.tap_sums_bn(geo, got, y, st, coef, got, S_fused, ws)
    torch.cuda.synchronize()
    assert torch.equal(got, want_dy)
    assert torch.equal(S_fused, S_sep)
