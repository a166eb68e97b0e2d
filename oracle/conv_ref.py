"""TEST INFRASTRUCTURE -- fp64 restatement of the dense contractions of createModel, one geometry at a time.

Conv3D / Conv2D after ZeroPadding ('valid', cross-correlation; model_training.py:192-193, 202-203), Conv2DTranspose
('same'; :246, :249, :252), Dense on the last axis (:184, :195) and the 1x1 heads (:254-255), with the
BatchNormalization(+ReLU) of the producing layer applied to the input first (:194, :204-206; padding stays zero because
ZeroPadding follows the activation in the graph), and the weight gradient of the same contraction (what fit() derives,
:299).  Written as a gather per kernel tap + one float64 matmul per tap so that the FULL Lyft layer geometries
(up to 320 000 positions x 27 taps x 64 x 64) take seconds on the host; tests/test_oracle_conv.py pins it against
torch.nn.functional.conv3d / conv2d / conv_transpose2d in float64 on small cases (parity with Keras itself stays
unpinned: TensorFlow is absent).

Layouts: activations (D, H, W, C); kernels (taps, Cin, Cout) with tap = (kd*KH + kh)*KW + kw -- the Keras
(kd,kh,kw,in,out) order flattened; a Conv2DTranspose kernel (kh,kw,out,in) must be transposed to that by the caller.
mode 0: src = o*stride - pad + k;  mode 1 (transposed): src = (o + pad - k)/stride where divisible and in range.
"""
import numpy as np


def _src_index(o_n, k, stride, pad, n_in, mode):
    """source coordinate and validity of every output coordinate 0..o_n-1 for tap offset k along one axis"""
    o = np.arange(o_n)
    if mode == 0:
        s = o * stride - pad + k
        return s, (s >= 0) & (s < n_in)
    t = o + pad - k
    s = t // stride
    return s, (t >= 0) & (t % stride == 0) & (s < n_in)


def transform_input(x, in_bn=None, relu=False):
    """f(x) = [relu](x*scale + shift) in float64; in_bn = (scale, shift) per channel or None."""
    x = np.asarray(x, dtype=np.float64)
    if in_bn is not None:
        x = x * np.asarray(in_bn[0], np.float64) + np.asarray(in_bn[1], np.float64)
    if relu:
        x = np.maximum(x, 0.0)
    return x


def _gathered(xf, out_dims, kernel, stride, pad, mode, rows=None):
    """yields (tap, A) with A (M, Cin) = f(x) gathered at src(m, tap), zero where the tap falls outside / does not divide.
    rows: optional (V, 3) integer output coordinates (d, h, w) -- a row list instead of every position of out_dims."""
    Di, Hi, Wi, Cin = xf.shape
    Do, Ho, Wo = out_dims
    KD, KH, KW = kernel
    for kd in range(KD):
        sd, vd = _src_index(Do, kd, stride[0], pad[0], Di, mode)
        if not vd.any():
            continue
        for kh in range(KH):
            sh, vh = _src_index(Ho, kh, stride[1], pad[1], Hi, mode)
            for kw in range(KW):
                sw, vw = _src_index(Wo, kw, stride[2], pad[2], Wi, mode)
                tap = (kd * KH + kh) * KW + kw
                if rows is not None:
                    r = np.asarray(rows)
                    valid = vd[r[:, 0]] & vh[r[:, 1]] & vw[r[:, 2]]
                    A = xf[np.clip(sd, 0, Di - 1)[r[:, 0]], np.clip(sh, 0, Hi - 1)[r[:, 1]], np.clip(sw, 0, Wi - 1)[r[:, 2]]]
                    yield tap, np.where(valid[:, None], A, 0.0)
                    continue
                valid = vd[:, None, None] & vh[None, :, None] & vw[None, None, :]
                A = xf[np.clip(sd, 0, Di - 1)[:, None, None], np.clip(sh, 0, Hi - 1)[None, :, None],
                       np.clip(sw, 0, Wi - 1)[None, None, :]]
                A = np.where(valid[..., None], A, 0.0).reshape(-1, Cin)
                yield tap, A


def conv_forward(x, W, out_dims, kernel, stride, pad, mode=0, bias=None, in_bn=None, relu=False, rows=None):
    """out (Do, Ho, Wo, Cout) float64 -- (V, Cout) for a row list.  x (Di,Hi,Wi,Cin), W (taps, Cin, Cout)."""
    xf = transform_input(x, in_bn, relu)
    W = np.asarray(W, dtype=np.float64)
    M = int(np.prod(out_dims)) if rows is None else len(rows)
    out = np.zeros((M, W.shape[2]), dtype=np.float64)
    for tap, A in _gathered(xf, out_dims, kernel, stride, pad, mode, rows):
        out += A @ W[tap]
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out.reshape(*out_dims, W.shape[2]) if rows is None else out


def conv_wgrad(x, dy, out_dims, kernel, stride, pad, mode=0, in_bn=None, relu=False, rows=None):
    """dW (taps, Cin, Cout) float64 = sum_m f(x)[src(m, tap)]^T dy[m].  dy (Do,Ho,Wo,Cout), or (V, Cout) with a row list."""
    xf = transform_input(x, in_bn, relu)
    dyf = np.asarray(dy, dtype=np.float64).reshape(-1, dy.shape[-1])
    taps = kernel[0] * kernel[1] * kernel[2]
    dW = np.zeros((taps, xf.shape[-1], dyf.shape[1]), dtype=np.float64)
    for tap, A in _gathered(xf, out_dims, kernel, stride, pad, mode, rows):
        dW[tap] = A.T @ dyf
    return dW
