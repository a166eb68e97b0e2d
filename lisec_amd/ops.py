"""Thin Python wrappers over the dense-contraction entry points of the C ABI (include/lisec_hip.h
section 3).  Tensors are torch CUDA tensors used purely as device memory."""
import ctypes

import torch

from . import _lib
from ._lib import ConvGeom

IN_RELU, OUT_RELU, ACCUMULATE = 1, 2, 4


def geom(mode, in_dims, out_dims, kernel, stride, pad, cin, cout, in_stride=None, out_stride=None):
    """in_dims/out_dims/kernel/stride/pad: 3-tuples (d, h, w)."""
    g = ConvGeom()
    g.mode = mode
    g.Di, g.Hi, g.Wi = in_dims
    g.Do, g.Ho, g.Wo = out_dims
    g.KD, g.KH, g.KW = kernel
    g.sd, g.sh, g.sw = stride
    g.pd, g.ph, g.pw = pad
    g.Cin, g.Cout = cin, cout
    g.in_stride = in_stride if in_stride is not None else cin
    g.out_stride = out_stride if out_stride is not None else cout
    return g


def packed_floats(ntaps, K, N):
    return _lib.load().lisec_conv_packed_floats(ntaps, K, N)


def pack_weights(src, ntaps, K, N, tap_stride, k_stride, n_stride, out=None):
    """Repack a Keras-layout kernel (any strides) into the [tap][K/4][N][4] layout of the kernels."""
    lib = _lib.load()
    n = packed_floats(ntaps, K, N)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=src.device)
    assert out.numel() >= n
    _lib.check(lib.lisec_conv_pack_weights(_lib.ptr(src), ntaps, K, N, tap_stride, k_stride, n_stride,
                                           _lib.ptr(out), _lib.current_stream()))
    return out


def num_mblocks(g):
    n = _lib.load().lisec_conv_num_mblocks(ctypes.byref(g))
    if n < 0:
        raise _lib.LisecError(_lib.load().lisec_last_error().decode())
    return n


def conv_forward(g, x, wp, out, bias=None, in_bn=None, flags=0, stats=None):
    _lib.check(_lib.load().lisec_conv_forward(ctypes.byref(g), _lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias),
                                              _lib.ptr(in_bn), flags, _lib.ptr(out), _lib.ptr(stats),
                                              _lib.current_stream()))
    return out


def bn_finalize(partials, nparts, C, n_rows, gamma, beta, moving_mean, moving_var, unbiased, bnstate):
    _lib.check(_lib.load().lisec_bn_finalize(_lib.ptr(partials), nparts, C, float(n_rows), _lib.ptr(gamma),
                                             _lib.ptr(beta), _lib.ptr(moving_mean), _lib.ptr(moving_var),
                                             1 if unbiased else 0, _lib.ptr(bnstate), _lib.current_stream()))


def bn_fold(gamma, beta, moving_mean, moving_var, C, bnstate):
    _lib.check(_lib.load().lisec_bn_fold(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(moving_mean),
                                         _lib.ptr(moving_var), C, _lib.ptr(bnstate), _lib.current_stream()))
