"""MFMA implicit-GEMM kernels (C ABI) vs plain PyTorch fp32 CPU references of the same op.
Tolerance rtol 1e-4 + atol 1e-4*max|ref| (fp32 in/accumulate on both sides; only summation order differs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, rtol=1e-4):
    got, ref = got.double().cpu(), ref.double().cpu()
    atol = rtol * ref.abs().max().item()
    err = (got - ref).abs()
    assert (err <= atol + rtol * ref.abs()).all(), f"max err {err.max().item():.3e}, atol {atol:.3e}"


def _conv_case(D, H, W, Cin, Cout, k, stride, pad, in_bn=False, in_relu=False, out_relu=False, bias=True,
               stats=False, seed=0, splitk=True):
    from lisec_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(D, H, W, Cin, generator=g)
    w = torch.randn(*k, Cin, Cout, generator=g) * 0.1                   # Keras (kd,kh,kw,in,out)
    b = torch.randn(Cout, generator=g) if bias else None
    Do = (D + 2 * pad[0] - k[0]) // stride[0] + 1
    Ho = (H + 2 * pad[1] - k[1]) // stride[1] + 1
    Wo = (W + 2 * pad[2] - k[2]) // stride[2] + 1
    xin = x
    bn = None
    if in_bn:
        sc, sh = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g)
        bn = torch.cat([sc, sh, torch.zeros(2 * Cin)])
        xin = x * sc + sh
    if in_relu:
        xin = F.relu(xin)
    ref = F.conv3d(xin.permute(3, 0, 1, 2)[None], w.permute(4, 3, 0, 1, 2), b, stride=stride, padding=pad)[0]
    ref = ref.permute(1, 2, 3, 0)
    if out_relu:
        ref = F.relu(ref)
    geo = ops.geom(0, (D, H, W), (Do, Ho, Wo), k, stride, pad, Cin, Cout)
    wd = w.to(DEV)
    ntaps = k[0] * k[1] * k[2]
    wp = ops.pack_weights(wd, ntaps, Cin, Cout, Cin * Cout, Cout, 1)
    out = torch.full((Do, Ho, Wo, Cout), float("nan"), device=DEV)
    st = None
    if stats:
        st = torch.zeros(ops.num_mblocks(geo), 2, Cout, dtype=torch.float64, device=DEV)
    flags = (ops.IN_RELU if in_relu else 0) | (ops.OUT_RELU if out_relu else 0)
    ops.conv_forward(geo, x.to(DEV), wp, out, bias=None if b is None else b.to(DEV),
                     in_bn=None if bn is None else bn.to(DEV), flags=flags, stats=st, splitk=splitk)
    _close(out, ref)
    if stats:
        s = st.sum(0).cpu()
        flat = ref.reshape(-1, Cout).double()
        _close(s[0], flat.sum(0), rtol=1e-5)
        _close(s[1], (flat ** 2).sum(0), rtol=1e-5)


def test_conv3d_mid_layers():
    # addConv3DLayer geometries (model_training.py:236-238) on a small H, W
    _conv_case(8, 12, 20, 64, 64, (3, 3, 3), (2, 1, 1), (1, 1, 1), stats=True)
    _conv_case(4, 12, 20, 64, 64, (3, 3, 3), (1, 1, 1), (0, 1, 1), stats=True, seed=1)
    _conv_case(2, 9, 21, 64, 64, (3, 3, 3), (2, 1, 1), (1, 1, 1), seed=2)          # ragged M (189)


def test_conv2d_rpn_layers():
    # addConv2DLayer (model_training.py:201-207): BN+ReLU of the producer applied on load
    _conv_case(1, 24, 40, 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), stats=True)
    _conv_case(1, 12, 20, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), in_bn=True, in_relu=True, stats=True, seed=3)
    _conv_case(1, 7, 13, 128, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), in_bn=True, in_relu=True, seed=4)
    _conv_case(1, 5, 7, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), in_bn=True, in_relu=True, seed=5)


def test_dense_and_heads():
    # Dense(64, relu, no bias) on BN output (:195) and the 1x1 heads on the 768-channel concat (:254-255)
    _conv_case(2, 10, 30, 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), in_bn=True, out_relu=True, bias=False, seed=6)
    _conv_case(1, 10, 30, 768, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), seed=7)


@pytest.mark.parametrize("k,s,cin", [(3, 1, 128), (2, 2, 128), (4, 4, 256)])
def test_conv2d_transpose(k, s, cin):
    # Conv2DTranspose(256, padding='same') (:246,:249,:252) written into a 768-channel concat slice
    from lisec_amd import ops
    g = torch.Generator().manual_seed(k)
    H, W, cout = 6, 10, 256
    x = torch.randn(H, W, cin, generator=g)
    w = torch.randn(k, k, cout, cin, generator=g) * 0.1                 # Keras (kh,kw,out,in)
    b = torch.randn(cout, generator=g)
    sc, sh = torch.randn(cin, generator=g), torch.randn(cin, generator=g)
    xin = F.relu(x * sc + sh)
    pad = (k - s) // 2
    ref = F.conv_transpose2d(xin.permute(2, 0, 1)[None], w.permute(3, 2, 0, 1), b, stride=s, padding=pad)[0]
    ref = ref.permute(1, 2, 0)
    Ho, Wo = H * s, W * s
    assert ref.shape == (Ho, Wo, cout)
    geo = ops.geom(1, (1, H, W), (1, Ho, Wo), (1, k, k), (1, s, s), (0, pad, pad), cin, cout, out_stride=768)
    wp = ops.pack_weights(w.to(DEV), k * k, cin, cout, cout * cin, 1, cin)   # K = in (stride 1), N = out
    cat = torch.zeros(Ho, Wo, 768, device=DEV)
    bn = torch.cat([sc, sh, torch.zeros(2 * cin)]).to(DEV)
    ops.conv_forward(geo, x.to(DEV), wp, cat[:, :, 256:], bias=b.to(DEV), in_bn=bn, flags=ops.IN_RELU)
    _close(cat[:, :, 256:512], ref)
    assert (cat[:, :, :256] == 0).all() and (cat[:, :, 512:] == 0).all()
    if k == s:
        # kernel == stride: the same layer as a 1x1 GEMM with a pixel-shuffle store (the path LisecNet uses)
        geo = ops.geom(0, (1, H, W), (1, H, W), (1, 1, 1), (1, 1, 1), (0, 0, 0), cin, k * k * cout,
                       out_stride=768, ps=s, ps_channels=cout)
        wp = ops.pack_weights(w.to(DEV), 1, cin, k * k * cout, 0, 1, cin)
        cat2 = torch.zeros(Ho, Wo, 768, device=DEV)
        ops.conv_forward(geo, x.to(DEV), wp, cat2[:, :, 512:], bias=b.to(DEV), in_bn=bn, flags=ops.IN_RELU)
        _close(cat2[:, :, 512:], ref)
        assert (cat2[:, :, :512] == 0).all()


def test_accumulate_and_data_gradient():
    # data gradient of a stride-2 conv = mode-1 gather with (out,in)-transposed weights; ACCUMULATE adds
    from lisec_amd import ops
    g = torch.Generator().manual_seed(11)
    D, H, W, Cin, Cout = 4, 10, 14, 64, 64
    k, stride, pad = (3, 3, 3), (2, 1, 1), (1, 1, 1)
    x = torch.randn(D, H, W, Cin, generator=g, requires_grad=True)
    w = torch.randn(*k, Cin, Cout, generator=g) * 0.1
    y = F.conv3d(x.permute(3, 0, 1, 2)[None], w.permute(4, 3, 0, 1, 2), None, stride=stride, padding=pad)[0]
    y = y.permute(1, 2, 3, 0)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Do, Ho, Wo = y.shape[:3]
    geo = ops.geom(1, (Do, Ho, Wo), (D, H, W), k, stride, pad, Cout, Cin)
    wp = ops.pack_weights(w.to(DEV), 27, Cout, Cin, Cin * Cout, 1, Cout)       # K = out, N = in
    base = torch.randn(D, H, W, Cin, generator=g)
    dx = base.to(DEV).clone()
    ops.conv_forward(geo, dy.to(DEV), wp, dx, flags=ops.ACCUMULATE, splitk=False)
    _close(dx, x.grad + base)


@pytest.mark.parametrize("shape", [(4, 10, 14), (1, 25, 50)])
def test_output_mask_gates_the_stored_gradient(shape):
    """lisec_conv_forward_masked: stored value = mask > 0 ? value : 0 -- the ReLU gate of a Dense(relu) consumer folded
    into the data-gradient store (single pass and the K-sliced path, whose combine kernel applies the gate)."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(5)
    D, H, W = shape
    Cin = Cout = 64
    k, stride, pad = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    w = torch.randn(*k, Cin, Cout, generator=g) * 0.1
    dy = torch.randn(D, H, W, Cout, generator=g).to(DEV)
    mask = torch.randn(D, H, W, Cin, generator=g)
    mask[mask.abs() < 0.2] = 0.0                                   # exact zeros gate too (mask > 0 is strict)
    geo = ops.geom(1, (D, H, W), (D, H, W), k, stride, pad, Cout, Cin)
    wp = ops.pack_weights(w.to(DEV), 27, Cout, Cin, Cin * Cout, 1, Cout)
    for splitk in (True, False):
        plain = torch.empty(D, H, W, Cin, device=DEV)
        gated = torch.full((D, H, W, Cin), float("nan"), device=DEV)
        ops.conv_forward(geo, dy, wp, plain, splitk=splitk)
        ops.conv_forward(geo, dy, wp, gated, splitk=splitk, out_mask=mask.to(DEV))
        assert torch.equal(gated, torch.where(mask.to(DEV) > 0, plain, torch.zeros_like(plain)))


@pytest.mark.parametrize("case", [
    dict(D=3, H=5, W=130, Cin=64, Cout=64, k=(3, 3, 3), stride=(2, 1, 1), pad=(1, 1, 1)),          # mid1-like, tiles cross lines
    dict(D=4, H=3, W=400, Cin=64, Cout=64, k=(3, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1)),          # mid2-like
    dict(D=1, H=7, W=200, Cin=128, Cout=128, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1), in_bn=True, in_relu=True),
    dict(D=1, H=9, W=126, Cin=64, Cout=192, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1), out_relu=True),
    dict(D=2, H=4, W=257, Cin=80, Cout=64, k=(3, 3, 3), stride=(2, 1, 1), pad=(1, 1, 1), in_bn=True),   # ragged channel slab
    dict(D=1, H=1, W=126, Cin=64, Cout=16, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1)),          # one partial tile, 16 columns
    dict(D=1, H=3, W=127, Cin=64, Cout=64, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1), bias=False),
    dict(D=1, H=3, W=128, Cin=64, Cout=64, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1)),          # tiles == lines
    dict(D=3, H=2, W=129, Cin=64, Cout=64, k=(3, 3, 3), stride=(1, 1, 1), pad=(1, 1, 1)),
    dict(D=1, H=50, W=100, Cin=128, Cout=128, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1), in_bn=True, in_relu=True),  # 3 lines / tile
    dict(D=1, H=25, W=50, Cin=256, Cout=256, k=(1, 3, 3), stride=(1, 1, 1), pad=(0, 1, 1), in_bn=True, in_relu=True),   # narrower than 64: generic kernel
    dict(D=2, H=7, W=43, Cin=64, Cout=64, k=(3, 3, 3), stride=(1, 1, 1), pad=(1, 1, 1)),                                # generic kernel
])
@pytest.mark.parametrize("splitk", [False, True])
def test_w_halo_kernel_forward(case, splitk):
    """Geometries served by k_igemm_halo (3 taps, stride 1, pad 1 along w, Wo >= 126): one A tile per (kd, kh, slab).
    splitk=True lets the plan cut these small test layers into K slices of whole A tiles + the combine kernel."""
    c = dict(case)
    _conv_case(c.pop("D"), c.pop("H"), c.pop("W"), c.pop("Cin"), c.pop("Cout"), c.pop("k"), c.pop("stride"), c.pop("pad"),
               stats=True, splitk=splitk, **c)


@pytest.mark.parametrize("splitk", [False, True])
@pytest.mark.parametrize("W", [200, 100])
@pytest.mark.parametrize("stride,pad,D", [((1, 1, 1), (0, 1, 1), 4), ((2, 1, 1), (1, 1, 1), 4), ((1, 1, 1), (1, 1, 1), 1)])
def test_w_halo_kernel_data_gradient(stride, pad, D, splitk, W):
    """mode 1 through the halo kernel: fragment base moves by 2 - kw; the depth stride keeps its divisibility rule."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(3)
    H, Cin, Cout = 3 if W == 200 else 11, 64, 64
    k = (3, 3, 3) if D > 1 else (1, 3, 3)
    pad = pad if D > 1 else (0, 1, 1)
    x = torch.randn(D, H, W, Cin, generator=g, requires_grad=True)
    w = torch.randn(*k, Cin, Cout, generator=g) * 0.1
    y = F.conv3d(x.permute(3, 0, 1, 2)[None], w.permute(4, 3, 0, 1, 2), None, stride=stride, padding=pad)[0]
    y = y.permute(1, 2, 3, 0)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Do, Ho, Wo = y.shape[:3]
    geo = ops.geom(1, (Do, Ho, Wo), (D, H, W), k, stride, pad, Cout, Cin)
    ntaps = k[0] * k[1] * k[2]
    wp = ops.pack_weights(w.to(DEV), ntaps, Cout, Cin, Cin * Cout, 1, Cout)
    base = torch.randn(D, H, W, Cin, generator=g)
    dx = base.to(DEV).clone()
    ops.conv_forward(geo, dy.to(DEV), wp, dx, flags=ops.ACCUMULATE, splitk=splitk)
    _close(dx, x.grad + base)


@pytest.mark.parametrize("splitk", [False, True])
@pytest.mark.parametrize("H,W,Cin,Cout", [(10, 64, 64, 128), (6, 200, 128, 128), (50, 100, 128, 256)])
def test_stride2_data_gradient_parity_classes(H, W, Cin, Cout, splitk):
    """Data gradient of a stride-2 Conv2D (mode 1, stride 2 along h and w): rows are visited in parity classes so that
    each tile only runs the taps that divide; results land at the true positions (also through the K-sliced combine)."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(13)
    k, stride, pad = (1, 3, 3), (1, 2, 2), (0, 1, 1)
    x = torch.randn(1, H, W, Cin, generator=g, requires_grad=True)
    w = torch.randn(*k, Cin, Cout, generator=g) * 0.1
    y = F.conv3d(x.permute(3, 0, 1, 2)[None], w.permute(4, 3, 0, 1, 2), None, stride=stride, padding=pad)[0]
    y = y.permute(1, 2, 3, 0)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Do, Ho, Wo = y.shape[:3]
    geo = ops.geom(1, (Do, Ho, Wo), (1, H, W), k, stride, pad, Cout, Cin)
    wp = ops.pack_weights(w.to(DEV), 9, Cout, Cin, Cin * Cout, 1, Cout)
    base = torch.randn(1, H, W, Cin, generator=g)
    mask = torch.randn(1, H, W, Cin, generator=g)
    dx = base.to(DEV).clone()
    ops.conv_forward(geo, dy.to(DEV), wp, dx, flags=ops.ACCUMULATE, splitk=splitk)
    _close(dx, x.grad + base)
    dxm = torch.full((1, H, W, Cin), float("nan"), device=DEV)
    ops.conv_forward(geo, dy.to(DEV), wp, dxm, splitk=splitk, out_mask=mask.to(DEV))
    _close(dxm, torch.where(mask > 0, x.grad.detach(), torch.zeros_like(mask)))


@pytest.mark.parametrize("splitk", [False, True])
@pytest.mark.parametrize("mode,stride,W", [(1, (1, 1, 1), 130), (1, (1, 2, 2), 64), (0, (1, 2, 2), 40)])
def test_backward_statistics_folded_into_the_store(mode, stride, W, splitk):
    """lisec_conv_forward_ex with bwd_y: the per-tile partials are (sum dz, sum dz*yhat) of the gradient that was just
    stored (pass 1 of lisec_bn_backward), also in the parity-class row order and through the K-sliced combine; then
    lisec_bn_backward_apply must equal the three-pass lisec_bn_backward on the same gradient."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(17)
    H, Cg, Cs = 6, 64, 128                      # Cg: channels of the gradient that is produced, Cs: of its source
    k, pad = (1, 3, 3), (0, 1, 1)
    if mode == 1:                               # data gradient of a conv: source positions = that conv's output map
        Hs, Ws = (H + 2 - 3) // stride[1] + 1, (W + 2 - 3) // stride[2] + 1
        geo = ops.geom(1, (1, Hs, Ws), (1, H, W), k, stride, pad, Cs, Cg)
    else:                                       # data gradient of a stride-s deconv: a strided conv over its output
        Hs, Ws = H * stride[1], W * stride[2]
        k, pad = (1, stride[1], stride[2]), (0, 0, 0)
        geo = ops.geom(0, (1, Hs, Ws), (1, H, W), k, stride, pad, Cs, Cg)
    ntaps = k[1] * k[2]
    src = torch.randn(1, Hs, Ws, Cs, generator=g).to(DEV)
    w = (torch.randn(ntaps, Cs, Cg, generator=g) * 0.1).to(DEV)
    wp = ops.pack_weights(w, ntaps, Cs, Cg, Cs * Cg, Cg, 1)
    y = (torch.randn(H * W, Cg, generator=g) * 2 + 0.3).to(DEV)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = torch.rsqrt(var + 1e-3)
    gamma, beta = torch.rand(Cg, generator=g).to(DEV) + 0.5, torch.randn(Cg, generator=g).to(DEV)
    st = torch.cat([gamma * inv, beta - mean * gamma * inv, mean, inv]).contiguous()
    M = H * W
    for relu in (True, False):
        plain = torch.empty(1, H, W, Cg, device=DEV)
        ops.conv_forward(geo, src, wp, plain, splitk=splitk)
        dg0, db0 = torch.empty(Cg, device=DEV), torch.empty(Cg, device=DEV)
        want = torch.empty(M, Cg, device=DEV)
        ops.bn_backward(plain, Cg, y, st, M, Cg, relu, dg0, db0, want)
        nparts = ops.num_mblocks_bwd(geo)
        parts = torch.full((nparts, 2, Cg), float("nan"), dtype=torch.float64, device=DEV)
        got = torch.empty(1, H, W, Cg, device=DEV)
        ops.conv_forward(geo, src, wp, got, splitk=splitk, bwd=(y, st, relu), stats=parts)
        # same contraction, possibly another K-slice order (a call without a partial table may keep its slices inside
        # the workgroups): equal to fp32 summation noise
        _close(got, plain, rtol=1e-5)
        dz = got.reshape(M, Cg).double()
        if relu:
            dz = dz * ((y.double() * st[:Cg].double() + st[Cg:2 * Cg].double()) > 0)
        yhat = (y.double() - mean.double()) * inv.double()
        tot = parts.sum(0)
        _close(tot[0], dz.sum(0), rtol=1e-5)
        _close(tot[1], (dz * yhat).sum(0), rtol=1e-5)
        dg1, db1 = torch.empty(Cg, device=DEV), torch.empty(Cg, device=DEV)
        ops.bn_backward_apply(got, Cg, y, st, M, Cg, relu, parts, nparts, dg1, db1, got)
        _close(got.reshape(M, Cg), want, rtol=1e-5)
        _close(dg1, dg0, rtol=1e-5)
        _close(db1, db0, rtol=1e-5)


@pytest.mark.parametrize("count,capacity", [(700, 1000), (60000, 100000)])
def test_row_list_data_gradient_and_the_tile_queue(count, capacity):
    """lisec_conv_forward over a ROW LIST (the data gradient of the first Conv3D at the occupied cells only): rows of the
    dense data gradient picked at the listed positions; rows beyond the device-side count stay untouched.  The big list
    (>= 768 tiles of capacity) also runs with lisec_conv_extras.queue -- resident workgroups drawing tiles from a
    counter -- which must give the same bits and leave the counter words zero."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(29)
    C = 64
    ind, outd, k, s, p = (8, 12, 20), (4, 12, 20), (3, 3, 3), (2, 1, 1), (1, 1, 1)
    dg = ops.geom(1, outd, ind, k, s, p, C, C)                      # gradient wrt the conv's input: gathers from dz
    dz = torch.randn(*outd, C, generator=g).to(DEV)
    w = (torch.randn(27, C, C, generator=g) * 0.1).to(DEV)          # (tap, Cin, Cout) of the forward layer
    wt = ops.pack_weights(w, 27, C, C, C * C, 1, C)                 # transposed: K = Cout, N = Cin
    dense = torch.empty(*ind, C, device=DEV)
    ops.conv_forward(dg, dz, wt, dense)
    cells = torch.randint(0, ind[0] * ind[1] * ind[2], (count,), generator=g).sort().values
    coords = torch.stack([cells // (ind[1] * ind[2]), (cells // ind[2]) % ind[1], cells % ind[2]], 1).int()
    coords_dev = torch.zeros(capacity, 3, dtype=torch.int32, device=DEV)
    coords_dev[:count] = coords.to(DEV)
    n = torch.tensor([count], dtype=torch.int32, device=DEV)
    want = dense.reshape(-1, C)[cells.to(DEV)]
    out = torch.full((capacity, C), 7.0, device=DEV)
    ops.conv_forward(dg, dz, wt, out, rows=(coords_dev, n, capacity))
    _close(out[:count], want, rtol=1e-5)
    assert (out[count:] == 7.0).all()
    queue = torch.zeros(2, dtype=torch.int32, device=DEV)
    for _ in range(2):                                              # twice: the first call must leave the counter zero
        out_q = torch.full((capacity, C), 7.0, device=DEV)
        ops.conv_forward(dg, dz, wt, out_q, rows=(coords_dev, n, capacity), queue=queue)
        torch.cuda.synchronize()
        assert (queue == 0).all()
        if capacity >= 768 * 128:
            assert torch.equal(out_q, out)                          # same tiles, same arithmetic
        else:
            _close(out_q[:count], want, rtol=1e-5)


def test_dense64_resident_workgroups_forward_and_data_gradient():
    """Dense(64) over enough rows (>= 768 tiles) for the resident-workgroup kernel (k_dense64): forward with the
    BatchNormalization of the producer applied on load, bias and ReLU (model_training.py:195), against a plain fp32
    PyTorch evaluation; data gradient with the BatchNormalization-backward sums in a lisec_bn_sink against their
    definitions ((sum dz, sum dz*yhat) -> dgamma, dbeta, coef), twice (the sink must be left ready for the next call)."""
    from lisec_amd import ops
    g = torch.Generator().manual_seed(31)
    dims, C = (4, 161, 153), 64                       # 98 532 rows: 770 tiles, a ragged last one
    M = dims[0] * dims[1] * dims[2]
    geo = ops.geom(0, dims, dims, (1, 1, 1), (1, 1, 1), (0, 0, 0), C, C)
    x = torch.randn(M, C, generator=g)
    w = torch.randn(C, C, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    sc, sh = torch.randn(C, generator=g), torch.randn(C, generator=g)
    wp = ops.pack_weights(w.to(DEV), 1, C, C, 0, C, 1)
    out = torch.full((M, C), float("nan"), device=DEV)
    ops.conv_forward(geo, x.to(DEV), wp, out, bias=b.to(DEV), in_bn=torch.cat([sc, sh, torch.zeros(2 * C)]).to(DEV),
                     flags=ops.OUT_RELU)
    _close(out, F.relu((x * sc + sh) @ w + b))
    # data gradient dz = du @ W^T with the statistics of the BatchNormalization it is about to cross
    du = torch.randn(M, C, generator=g)
    y = torch.randn(M, C, generator=g) * 1.5 + 0.3
    mean, var = y.mean(0), y.var(0, unbiased=False)
    inv = torch.rsqrt(var + 1e-3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    st = torch.cat([gamma * inv, beta - mean * gamma * inv, mean, inv]).to(DEV)
    wt = ops.pack_weights(w.to(DEV), 1, C, C, 0, 1, C)                 # transposed: K = out channel, N = in channel
    dgamma, dbeta = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    sink = ops.BnSink(C, M, DEV, dgamma=dgamma, dbeta=dbeta)
    want = du.double() @ w.double().t()
    yhat = (y.double() - mean.double()) * inv.double()
    for _ in range(2):
        dz = torch.full((M, C), float("nan"), device=DEV)
        ops.conv_forward(geo, du.to(DEV), wt, dz, bwd=(y.to(DEV), st, False), sink=sink)
        torch.cuda.synchronize()
        _close(dz, want.float())
        _close(dbeta, want.sum(0), rtol=1e-5)
        _close(dgamma, (want * yhat).sum(0), rtol=1e-5)
        _close(sink.coef[:C], want.mean(0), rtol=1e-5)
        _close(sink.coef[C:], (want * yhat).mean(0), rtol=1e-5)
        assert (sink.acc == 0).all()
