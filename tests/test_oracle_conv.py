"""oracle/conv_ref.py (the fp64 tap-gather restatement used at the full Lyft layer geometries by
tests/test_gpu_lyft_layers.py) against torch.nn.functional conv3d / conv2d / conv_transpose2d in float64 on small cases:
Conv3D after ZeroPadding3D (model_training.py:192-193), Conv2D after ZeroPadding2D (:202-203), Conv2DTranspose 'same'
(:246,:249,:252), and their weight gradients by autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv_ref

CASES = [
    # name, in dims, out dims, kernel, stride, pad, cin, cout   (mode 0: ZeroPadding + 'valid')
    ("mid1-like s(2,1,1) p(1,1,1)", (8, 6, 10), (4, 6, 10), (3, 3, 3), (2, 1, 1), (1, 1, 1), 8, 12),
    ("mid2-like s(1,1,1) p(0,1,1)", (4, 6, 10), (2, 6, 10), (3, 3, 3), (1, 1, 1), (0, 1, 1), 8, 12),
    ("rpn conv0 s2", (1, 12, 20), (1, 6, 10), (1, 3, 3), (1, 2, 2), (0, 1, 1), 8, 16),
    ("rpn conv1 s1", (1, 6, 10), (1, 6, 10), (1, 3, 3), (1, 1, 1), (0, 1, 1), 16, 16),
    ("dense 1x1", (2, 5, 7), (2, 5, 7), (1, 1, 1), (1, 1, 1), (0, 0, 0), 16, 12),
]


@pytest.mark.parametrize("name,ind,outd,k,s,p,cin,cout", CASES)
def test_conv_forward_and_wgrad_equal_torch(name, ind, outd, k, s, p, cin, cout):
    rng = np.random.default_rng(len(name))
    x = rng.normal(0, 1, (*ind, cin))
    W = rng.normal(0, 0.2, (k[0] * k[1] * k[2], cin, cout))
    b = rng.normal(0, 0.1, cout)
    scale, shift = rng.uniform(0.5, 1.5, cin), rng.normal(0, 0.3, cin)
    dy = rng.normal(0, 1, (*outd, cout))
    xt = torch.from_numpy(x).requires_grad_(False)
    a = torch.relu(xt * torch.from_numpy(scale) + torch.from_numpy(shift))          # BN + ReLU of the producer
    a = a.permute(3, 0, 1, 2)[None]                                                 # (1, C, D, H, W)
    a = F.pad(a, (p[2], p[2], p[1], p[1], p[0], p[0]))                              # ZeroPadding AFTER the activation
    wt = torch.from_numpy(W).reshape(*k, cin, cout).permute(4, 3, 0, 1, 2).contiguous().requires_grad_(True)
    y = F.conv3d(a, wt, torch.from_numpy(b), stride=s)
    ref = y[0].permute(1, 2, 3, 0).detach().numpy()
    got = conv_ref.conv_forward(x, W, outd, k, s, p, mode=0, bias=b, in_bn=(scale, shift), relu=True)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12
    (y * torch.from_numpy(dy).permute(3, 0, 1, 2)[None]).sum().backward()
    dW_ref = wt.grad.permute(2, 3, 4, 1, 0).reshape(-1, cin, cout).numpy()
    dW = conv_ref.conv_wgrad(x, dy, outd, k, s, p, mode=0, in_bn=(scale, shift), relu=True)
    assert np.abs(dW - dW_ref).max() < 1e-11
    # the data gradient of a mode-0 contraction is the mode-1 contraction of dy with the transposed kernel
    xg = torch.from_numpy(x).requires_grad_(True)
    a2 = F.pad(xg.permute(3, 0, 1, 2)[None], (p[2], p[2], p[1], p[1], p[0], p[0]))
    (F.conv3d(a2, wt.detach(), None, stride=s) * torch.from_numpy(dy).permute(3, 0, 1, 2)[None]).sum().backward()
    dx = conv_ref.conv_forward(dy, np.transpose(W, (0, 2, 1)), ind, k, s, p, mode=1)
    assert np.abs(dx - xg.grad.numpy()).max() < 1e-11


@pytest.mark.parametrize("k,s,hw", [(3, 1, (6, 10)), (2, 2, (3, 5)), (4, 4, (2, 3))])
def test_conv2d_transpose_same_equals_torch(k, s, hw):
    """Conv2DTranspose(256, k, strides=s, padding='same'): out = in * s; k3/s1 pads 1, kernel == stride pads 0."""
    rng = np.random.default_rng(10 * k + s)
    cin, cout = 8, 12
    h, w = hw
    pad = (k - s) // 2
    x = rng.normal(0, 1, (1, h, w, cin))
    Wk = rng.normal(0, 0.3, (k, k, cout, cin))                                      # Keras layout (kh, kw, out, in)
    W = np.transpose(Wk, (0, 1, 3, 2)).reshape(k * k, cin, cout)
    got = conv_ref.conv_forward(x, W, (1, h * s, w * s), (1, k, k), (1, s, s), (0, pad, pad), mode=1)
    xt = torch.from_numpy(x[0]).permute(2, 0, 1)[None]
    wt = torch.from_numpy(Wk).permute(3, 2, 0, 1).contiguous()                      # torch: (in, out, kh, kw)
    ref = F.conv_transpose2d(xt, wt, stride=s, padding=pad)[0].permute(1, 2, 0).numpy()
    assert got[0].shape == ref.shape and np.abs(got[0] - ref).max() < 1e-12
