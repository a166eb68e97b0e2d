"""Thin Python wrappers over the dense-contraction entry points of the C ABI (include/lisec_hip.h
section 3).  Tensors are torch CUDA tensors used purely as device memory."""
import ctypes

import torch

from . import _lib
from ._lib import ConvGeom

IN_RELU, OUT_RELU, ACCUMULATE, DY_RELU, TAG_ROOFLINE = 1, 2, 4, 8, 32


def geom(mode, in_dims, out_dims, kernel, stride, pad, cin, cout, in_stride=None, out_stride=None, ps=0,
         ps_channels=0):
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
    g.ps, g.ps_channels = ps, ps_channels
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


class PackTable:
    """Device-resident table for lisec_conv_pack_weights_batched: entries (src, dst, ntaps, K, N, strides)."""

    def __init__(self, entries, device):
        arr = (_lib.PackDesc * len(entries))()
        start = 0
        for d, (src, dst, ntaps, K, N, ts, ks, ns) in zip(arr, entries):
            Kp, Np = (K + 63) // 64 * 64, (N + 63) // 64 * 64
            d.src, d.dst = src.data_ptr(), dst.data_ptr()
            d.tap_stride, d.k_stride, d.n_stride, d.start = ts, ks, ns, start
            d.ntaps, d.K, d.N, d.Kp, d.Np = ntaps, K, N, Kp, Np
            assert dst.numel() >= ntaps * Kp * Np
            start += ntaps * Kp * Np
        self.total, self.n = start, len(entries)
        self.keep = [e[0] for e in entries] + [e[1] for e in entries]
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)

    def run(self):
        _lib.check(_lib.load().lisec_conv_pack_weights_batched(_lib.ptr(self.table), self.n, self.total,
                                                               _lib.current_stream()))


class CopyTable:
    """Device-resident table for lisec_copy2d_batched: entries (src, dst) of equal 1-D / 2-D shapes (views allowed)."""

    def __init__(self, pairs, device):
        arr = (_lib.CopyDesc * len(pairs))()
        for d, (src, dst) in zip(arr, pairs):
            assert src.shape == dst.shape and src.dim() in (1, 2) and src.dtype == dst.dtype == torch.float32
            rows, cols = (1, src.shape[0]) if src.dim() == 1 else src.shape
            assert src.stride(-1) == 1 and dst.stride(-1) == 1
            d.src, d.dst, d.rows, d.cols = src.data_ptr(), dst.data_ptr(), rows, cols
            d.src_stride = src.stride(0) if src.dim() == 2 else cols
            d.dst_stride = dst.stride(0) if dst.dim() == 2 else cols
        self.n, self.keep = len(pairs), pairs
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)

    def run(self):
        _lib.check(_lib.load().lisec_copy2d_batched(_lib.ptr(self.table), self.n, _lib.current_stream()))


def num_mblocks(g):
    n = _lib.load().lisec_conv_num_mblocks(ctypes.byref(g))
    if n < 0:
        raise _lib.LisecError(_lib.load().lisec_last_error().decode())
    return n


_SPLITK_WS = {}


def conv_workspace(g, device, row_capacity=0, tag="main"):
    """Shared split-K scratch for `g` (None when the layer is large enough to run in one pass); one buffer per stream
    role (tag): contractions that may run at the same time on two streams must not share their slabs.  Zero-filled:
    it starts with the arrival counters of the K slices, which every call leaves at zero (include/lisec_hip.h)."""
    lib = _lib.load()
    need = (lib.lisec_conv_forward_rows_workspace_bytes(ctypes.byref(g), row_capacity) if row_capacity > 0
            else lib.lisec_conv_forward_workspace_bytes(ctypes.byref(g)))
    if need == 0:
        return None
    key = (str(device), tag)
    if key not in _SPLITK_WS or _SPLITK_WS[key].numel() < need:
        if key in _SPLITK_WS:
            torch.cuda.synchronize(device)          # a call on another stream may still be using the old buffer
        _SPLITK_WS[key] = torch.zeros(need, dtype=torch.uint8, device=device)
        _lib.bump_alloc_generation()                # recorded step plans hold the old address
    return _SPLITK_WS[key]


class BnSink:
    """lisec_bn_sink: the per-tile BatchNormalization sums of a contraction go into fixed-point accumulators and the
    last workgroup of the call finalises them (no stats table, no finaliser launch).  Forward: bnstate (+ moving
    statistics); backward: dgamma / dbeta / coef (feed bn_backward_apply_coef)."""

    def __init__(self, C, n_rows, device, gamma=None, beta=None, moving_mean=None, moving_var=None, unbiased=True,
                 bnstate=None, dgamma=None, dbeta=None):
        lib = _lib.load()
        self.acc = torch.zeros(lib.lisec_bn_sink_words(C), dtype=torch.int64, device=device)    # zero before first use
        self.backward = dgamma is not None
        self.coef = torch.zeros(2 * C, dtype=torch.float32, device=device) if self.backward else None
        self.keep = (gamma, beta, moving_mean, moving_var, bnstate, dgamma, dbeta)
        d = _lib.BnSinkDesc()
        d.acc, d.kind, d.unbiased_moving, d.n_rows = self.acc.data_ptr(), 2 if self.backward else 1, 1 if unbiased else 0, float(n_rows)
        d.gamma, d.beta = _lib.ptr(gamma), _lib.ptr(beta)
        d.moving_mean, d.moving_var, d.bnstate = _lib.ptr(moving_mean), _lib.ptr(moving_var), _lib.ptr(bnstate)
        d.dgamma, d.dbeta, d.coef = _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(self.coef)
        self.desc = d
        self.ref = ctypes.pointer(d)


def _fold_fields(fold):
    if fold is None:
        return None, None, None, 0
    return _lib.ptr(fold[0]), _lib.ptr(fold[1]), _lib.ptr(fold[2]), 1 if fold[3] else 0


def conv_forward(g, x, wp, out, bias=None, in_bn=None, flags=0, stats=None, splitk=True, rows=None, out_mask=None,
                 bwd=None, sink=None, ws_tag="main", queue=None, tail=None, fold=None, dense_dw=None):
    """rows: optional (row_coords int32 (cap,3), row_count int32 device scalar, capacity) row list.
    out_mask: optional tensor laid out like `out`; values are stored as 0 where out_mask <= 0.
    bwd: optional (y, bnstate, relu): `out` is a gradient about to cross that BatchNormalization(+ReLU) backwards and
    `stats` (num_mblocks_bwd(g) rows) receives the per-tile (sum dz, sum dz*yhat) -- see bn_backward_apply.
    sink: optional BnSink taking the per-tile sums instead of `stats` (finalised inside the call).
    queue: optional int32[2] device tensor, zero before its first use: lets a big row list run as resident workgroups
    that draw their tiles from a counter (lisec_conv_extras.queue).
    tail: optional (packed 64 x 64 kernel, out2): out2 = out (as stored) @ kernel rides on the tile; bwd / sink then describe
    out2 (lisec_conv_extras.tail_w).
    fold: optional (y, bnstate, coef, relu): x is a gradient about to cross that BatchNormalization(+ReLU) backwards and the
    apply pass runs on load (lisec_conv_extras.in_y).
    dense_dw: optional float32 tensor of dense_dw_slabs() * 4096 elements: the call is a Dense(64)'s data gradient (bwd + sink)
    and also leaves the per-workgroup slabs of the Dense's weight gradient there (lisec_conv_extras.dense_dw; dense_dw_reduce)."""
    rc, rn, cap = rows if rows is not None else (None, None, 0)
    ws = conv_workspace(g, out.device, cap, ws_tag) if splitk else None
    ex = _lib.ConvExtras(_lib.ptr(out_mask), _lib.ptr(bwd[0]) if bwd is not None else None,
                         _lib.ptr(bwd[1]) if bwd is not None else None, 1 if (bwd is not None and bwd[2]) else 0,
                         sink.ref if sink is not None else None, _lib.ptr(queue),
                         _lib.ptr(tail[0]) if tail is not None else None, _lib.ptr(tail[1]) if tail is not None else None,
                         *_fold_fields(fold), _lib.ptr(dense_dw))
    _lib.check(_lib.load().lisec_conv_forward_ex(ctypes.byref(g), _lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias),
                                                 _lib.ptr(in_bn), flags, _lib.ptr(out), ctypes.byref(ex),
                                                 _lib.ptr(stats), _lib.ptr(ws),
                                                 ws.numel() if ws is not None else 0,
                                                 _lib.ptr(rc), _lib.ptr(rn), cap, _lib.current_stream()))
    return out


def dense_dw_slabs():
    """Slabs (of 64 x 64 floats) a conv_forward(..., dense_dw=) call writes (lisec_dense_dw_slabs)."""
    return _lib.load().lisec_dense_dw_slabs()


def dense_dw_reduce(slabs, dW):
    """dW (64, 64) = the slabs of a conv_forward(..., dense_dw=slabs) call summed in index order (lisec_dense_dw_reduce)."""
    _lib.check(_lib.load().lisec_dense_dw_reduce(_lib.ptr(slabs), _lib.ptr(dW), _lib.current_stream()))
    return dW


def winograd_packed_floats(KD, K, N):
    return _lib.load().lisec_conv_winograd_packed_floats(KD, K, N)


def pack_weights_winograd(src, KD, K, N, tap_stride, k_stride, n_stride, flip=False, out=None):
    """G g G^T of a (KD, 3, 3, ...) kernel in the Winograd kernel's LDS image order (lisec_conv_pack_weights_winograd);
    flip mirrors the (kh, kw) taps: the kernel a data gradient (mode 1) takes."""
    n = winograd_packed_floats(KD, K, N)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=src.device)
    assert out.numel() >= n
    _lib.check(_lib.load().lisec_conv_pack_weights_winograd(_lib.ptr(src), KD, K, N, tap_stride, k_stride, n_stride,
                                                            1 if flip else 0, _lib.ptr(out), _lib.current_stream()))
    return out


def _wino_extras(out_mask, bwd, sink, tail=None):
    return _lib.ConvExtras(_lib.ptr(out_mask), _lib.ptr(bwd[0]) if bwd is not None else None,
                           _lib.ptr(bwd[1]) if bwd is not None else None, 1 if (bwd is not None and bwd[2]) else 0,
                           sink.ref if sink is not None else None, None,
                           _lib.ptr(tail[0]) if tail is not None else None, _lib.ptr(tail[1]) if tail is not None else None,
                           None, None, None, 0)


def winograd_supported(g, in_bn=False, flags=0, out_mask=None, bwd=None, sink=None, tail=None):
    ex = _wino_extras(out_mask, bwd, sink, tail)
    return bool(_lib.load().lisec_conv_winograd_supported(ctypes.byref(g), 1 if in_bn else 0, flags, ctypes.byref(ex)))


def conv_forward_winograd(g, x, wu, out, bias=None, in_bn=None, flags=0, out_mask=None, bwd=None, sink=None, tail=None):
    """conv_forward(...) in the Winograd F(2x2, 3x3) form (lisec_conv_forward_winograd); wu from pack_weights_winograd.
    tail: optional (packed 64 x 64 kernel, out2) as in conv_forward: out2 = out (as stored) @ kernel; bwd / sink describe out2."""
    ex = _wino_extras(out_mask, bwd, sink, tail)
    _lib.check(_lib.load().lisec_conv_forward_winograd(ctypes.byref(g), _lib.ptr(x), _lib.ptr(wu), _lib.ptr(bias),
                                                       _lib.ptr(in_bn), flags, _lib.ptr(out), ctypes.byref(ex),
                                                       _lib.current_stream()))
    return out


def conv_plan(g, in_bn=False, flags=0, stats=False, splitk=True, rows_capacity=0, out_mask=None, bwd=None, sink=None,
              queue=None, tail=None, fold=None, dense_dw=None):
    """The launch plan conv_forward(...) with the same arguments runs (lisec_conv_plan_query), as a dict."""
    lib = _lib.load()
    ws_bytes = 0
    if splitk:
        ws_bytes = (lib.lisec_conv_forward_rows_workspace_bytes(ctypes.byref(g), rows_capacity) if rows_capacity > 0
                    else lib.lisec_conv_forward_workspace_bytes(ctypes.byref(g)))
    ex = _lib.ConvExtras(_lib.ptr(out_mask), _lib.ptr(bwd[0]) if bwd is not None else None,
                         _lib.ptr(bwd[1]) if bwd is not None else None, 1 if (bwd is not None and bwd[2]) else 0,
                         sink.ref if sink is not None else None, _lib.ptr(queue),
                         _lib.ptr(tail[0]) if tail is not None else None, _lib.ptr(tail[1]) if tail is not None else None,
                         *_fold_fields(fold), _lib.ptr(dense_dw))
    plan = _lib.ConvPlan()
    _lib.check(lib.lisec_conv_plan_query(ctypes.byref(g), 1 if in_bn else 0, flags, ctypes.byref(ex), 1 if stats else 0,
                                         ws_bytes, 1 if rows_capacity > 0 else 0, rows_capacity, ctypes.byref(plan)))
    d = {n: getattr(plan, n) for n, _ in _lib.ConvPlan._fields_}
    d["kernel"] = _lib.KERNEL_NAMES[d["kernel"]]
    return d


def num_mblocks_bwd(g):
    n = _lib.load().lisec_conv_num_mblocks_bwd(ctypes.byref(g))
    if n < 0:
        raise _lib.LisecError(_lib.load().lisec_last_error().decode())
    return n


def bn_backward_apply(dA, da_stride, y, bnstate, M, C, relu, parts, nparts, dgamma, dbeta, dy):
    """Passes 2-3 of bn_backward on partials written by conv_forward(..., bwd=...)."""
    ws = _ew_workspace(y.device)
    _lib.check(_lib.load().lisec_bn_backward_apply(_lib.ptr(dA), da_stride, _lib.ptr(y), _lib.ptr(bnstate), M, C,
                                                   1 if relu else 0, _lib.ptr(parts), nparts, _lib.ptr(dgamma),
                                                   _lib.ptr(dbeta), _lib.ptr(dy), _lib.ptr(ws), ws.numel(),
                                                   _lib.current_stream()))


def bn_backward_apply_coef(dA, da_stride, y, bnstate, M, C, relu, coef, dy):
    """The apply pass alone, with the coefficients a backward BnSink produced."""
    _lib.check(_lib.load().lisec_bn_backward_apply_coef(_lib.ptr(dA), da_stride, _lib.ptr(y), _lib.ptr(bnstate), M, C,
                                                        1 if relu else 0, _lib.ptr(coef), _lib.ptr(dy),
                                                        _lib.current_stream()))


def bn_finalize(partials, nparts, C, n_rows, gamma, beta, moving_mean, moving_var, unbiased, bnstate):
    _lib.check(_lib.load().lisec_bn_finalize(_lib.ptr(partials), nparts, C, float(n_rows), _lib.ptr(gamma),
                                             _lib.ptr(beta), _lib.ptr(moving_mean), _lib.ptr(moving_var),
                                             1 if unbiased else 0, _lib.ptr(bnstate), _lib.current_stream()))


def bn_fold(gamma, beta, moving_mean, moving_var, C, bnstate):
    _lib.check(_lib.load().lisec_bn_fold(_lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(moving_mean),
                                         _lib.ptr(moving_var), C, _lib.ptr(bnstate), _lib.current_stream()))


def wgrad_workspace_bytes(g, row_capacity=0):
    return _lib.load().lisec_conv_wgrad_workspace_bytes(ctypes.byref(g), row_capacity)


def wgrad_plan(g, flags=0, dy_bn=False, rows_capacity=0):
    """The launch plan conv_wgrad(...) runs for these arguments (lisec_conv_wgrad_plan_query), as a dict."""
    plan = _lib.WgradPlan()
    _lib.check(_lib.load().lisec_conv_wgrad_plan_query(ctypes.byref(g), flags, 1 if dy_bn else 0,
                                                       1 if rows_capacity > 0 else 0, rows_capacity, ctypes.byref(plan)))
    return {n: getattr(plan, n) for n, _ in _lib.WgradPlan._fields_}


class WgradBatch:
    """lisec_conv_wgrad_batched over fixed tensors: items = [(geom, x, dy, dW, in_bn or None, flags, transpose_out), ...]
    (at most 6).  The stride-1 convolutions of one RPN block share one launch."""

    def __init__(self, items):
        self.n = len(items)
        self.keep = items
        self.arr = (_lib.WgradItem * self.n)()
        for a, (g, x, dy, dW, in_bn, flags, transpose_out) in zip(self.arr, items):
            a.g = ctypes.pointer(g)
            a.in_, a.in_bnstate, a.flags = x.data_ptr(), _lib.ptr(in_bn), flags
            a.dy, a.transpose_out, a.dW = dy.data_ptr(), 1 if transpose_out else 0, dW.data_ptr()

    def workspace_bytes(self):
        return _lib.load().lisec_conv_wgrad_batched_workspace_bytes(self.arr, self.n)

    def run(self, workspace):
        _lib.check(_lib.load().lisec_conv_wgrad_batched(self.arr, self.n, _lib.ptr(workspace),
                                                        workspace.numel() * workspace.element_size(),
                                                        _lib.current_stream()))


def conv_wgrad(g, x, dy, dW, workspace, in_bn=None, flags=0, transpose_out=False, dy_bn=None, rows=None):
    rc, rn, cap = rows if rows is not None else (None, None, 0)
    _lib.check(_lib.load().lisec_conv_wgrad(ctypes.byref(g), _lib.ptr(x), _lib.ptr(in_bn), flags, _lib.ptr(dy),
                                            _lib.ptr(dy_bn), _lib.ptr(workspace),
                                            workspace.numel() * workspace.element_size(),
                                            1 if transpose_out else 0, _lib.ptr(dW), _lib.ptr(rc), _lib.ptr(rn), cap,
                                            _lib.current_stream()))
    return dW


def wgrad_winograd_supported(g):
    return bool(_lib.load().lisec_conv_wgrad_winograd_supported(ctypes.byref(g)))


def wgrad_winograd_workspace_bytes(g):
    return _lib.load().lisec_conv_wgrad_winograd_workspace_bytes(ctypes.byref(g))


def conv_wgrad_winograd(g, x, dy, dW, workspace):
    """conv_wgrad(g, x, dy, dW, ...) in the Winograd F(2x2, 3x3) form (lisec_conv_wgrad_winograd): 64 -> 64 Conv3D blocks."""
    _lib.check(_lib.load().lisec_conv_wgrad_winograd(ctypes.byref(g), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(workspace),
                                                     workspace.numel() * workspace.element_size(), _lib.ptr(dW),
                                                     _lib.current_stream()))
    return dW


_EW = {}


def _ew_workspace(device, tag="main"):
    """Scratch of the element-wise entry points; launches that may run at the same time on different streams must not
    share one (tag: one buffer per stream role)."""
    key = (str(device), tag)
    if key not in _EW:
        _EW[key] = torch.empty(_lib.load().lisec_eltwise_workspace_bytes(), dtype=torch.uint8, device=device)
    return _EW[key]


def bn_backward(dA, da_stride, y, bnstate, M, C, relu, dgamma, dbeta, dy, dbias=None):
    ws = _ew_workspace(y.device)
    _lib.check(_lib.load().lisec_bn_backward(_lib.ptr(dA), da_stride, _lib.ptr(y), _lib.ptr(bnstate), M, C,
                                             1 if relu else 0, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dbias),
                                             _lib.ptr(dy), _lib.ptr(ws), ws.numel(), _lib.current_stream()))


def relu_mask(grad, act):
    _lib.check(_lib.load().lisec_relu_mask(_lib.ptr(grad), _lib.ptr(act), grad.numel(), _lib.current_stream()))


def colsum(x, stride, M, C, out, ws_tag="main"):
    ws = _ew_workspace(x.device, ws_tag)
    _lib.check(_lib.load().lisec_colsum(_lib.ptr(x), stride, M, C, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                        _lib.current_stream()))


def rpn_loss(head, y_cls, y_reg, M, kind, dhead, loss_out, grad_scale=1.0):
    ws = _ew_workspace(head.device)
    _lib.check(_lib.load().lisec_rpn_loss(_lib.ptr(head), _lib.ptr(y_cls), _lib.ptr(y_reg), M, kind, grad_scale,
                                          _lib.ptr(dhead), _lib.ptr(loss_out), _lib.ptr(ws), ws.numel(),
                                          _lib.current_stream()))


def sgd_nesterov_step(theta, grad, velocity, lr_t, momentum):
    _lib.check(_lib.load().lisec_sgd_nesterov_step(_lib.ptr(theta), _lib.ptr(grad), _lib.ptr(velocity),
                                                   theta.numel(), lr_t, momentum, _lib.current_stream()))


def sgd_nesterov_step_dev(theta, grad, velocity, lr, decay, momentum, state, advance=True):
    """state: int64[2] device tensor {iterations, 0}; lr_t is derived on the device (graph-replayable).
    advance=False: a part of the variables ahead of the rest of the step (the iteration count is left alone)."""
    fn = _lib.load().lisec_sgd_nesterov_step_dev if advance else _lib.load().lisec_sgd_nesterov_step_dev_part
    _lib.check(fn(_lib.ptr(theta), _lib.ptr(grad), _lib.ptr(velocity), theta.numel(), float(lr), float(decay), momentum,
                  _lib.ptr(state), _lib.current_stream()))


def fold_depth(x, out, D, HW, C, inverse=False, mask=None):
    """(D,H,W,C) <-> (H,W,C*D) (Permute + Reshape of model_training.py:242-243); inverse: gradient, ReLU-gated by mask."""
    _lib.check(_lib.load().lisec_fold_depth(_lib.ptr(x), _lib.ptr(out), D, HW, C, 1 if inverse else 0, _lib.ptr(mask),
                                            _lib.current_stream()))
    return out


def scale_(x, s):
    _lib.check(_lib.load().lisec_scale(_lib.ptr(x), x.numel(), s, _lib.current_stream()))


def conv_field_forward_workspace_bytes(g, row_capacity):
    return _lib.load().lisec_conv_field_forward_workspace_bytes(ctypes.byref(g), row_capacity)


def conv_field_forward(g, vout, delta, sample, wp, out, workspace, bias=None, sink=None):
    """First Conv3D over the VFE's compact output (constant + voxel rows) instead of the dense grid; sample: the
    VoxelSample the VFE ran on; workspace: uint8 tensor of conv_field_forward_workspace_bytes(g, sample.cap)."""
    _lib.check(_lib.load().lisec_conv_field_forward(
        ctypes.byref(g), _lib.ptr(vout), _lib.ptr(delta), _lib.ptr(sample.info), _lib.ptr(sample.coords),
        _lib.ptr(sample.cell_voxel), sample.cap, _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(out),
        sink.ref if sink is not None else None, _lib.ptr(workspace), workspace.numel(), _lib.current_stream()))
    return out


def tap_sums(g, dy, S, workspace):
    _lib.check(_lib.load().lisec_conv_tap_sums(ctypes.byref(g), _lib.ptr(dy), _lib.ptr(S), _lib.ptr(workspace),
                                               workspace.numel() * workspace.element_size(), _lib.current_stream()))


def tap_sums_bn(g, dz, y, bnstate, coef, dy, S, workspace):
    """tap_sums fused with bn_backward_apply_coef(relu=False) in front of it: dy (may be dz) is written, S summed."""
    _lib.check(_lib.load().lisec_conv_tap_sums_bn(ctypes.byref(g), _lib.ptr(dz), _lib.ptr(y), _lib.ptr(bnstate),
                                                  _lib.ptr(coef), _lib.ptr(dy), _lib.ptr(S), _lib.ptr(workspace),
                                                  workspace.numel() * workspace.element_size(), _lib.current_stream()))


def tap_sums_finish(g, workspace, S):
    """Second half of tap_sums_bn(..., S=None): the per-line sums left in `workspace` summed into S."""
    _lib.check(_lib.load().lisec_conv_tap_sums_finish(ctypes.byref(g), _lib.ptr(workspace),
                                                      workspace.numel() * workspace.element_size(), _lib.ptr(S),
                                                      _lib.current_stream()))


def tap_sums_workspace_bytes(g):
    return _lib.load().lisec_conv_tap_sums_workspace_bytes(ctypes.byref(g))


def const_field_grads(W, S, cvec, ntaps, cin, cout, dW=None, g_all=None, cvec_row=None, cvec_row_max=0):
    _lib.check(_lib.load().lisec_const_field_grads(_lib.ptr(W), _lib.ptr(S), _lib.ptr(cvec), _lib.ptr(cvec_row),
                                                   cvec_row_max, ntaps, cin, cout, _lib.ptr(dW), _lib.ptr(g_all),
                                                   _lib.current_stream()))


def head_compose(up_kernel, up_bias, head_w, taps, cin, cup, Wc, tap_stride, c_stride, bias_in=None, bias_out=None):
    """Composite kernel of one upsampling branch and the heads (lisec_head_compose): Wc[tap*tap_stride + c*c_stride + j]."""
    _lib.check(_lib.load().lisec_head_compose(_lib.ptr(up_kernel), _lib.ptr(up_bias), _lib.ptr(head_w), taps, cin, cup,
                                              tap_stride, c_stride, _lib.ptr(Wc), _lib.ptr(bias_in), _lib.ptr(bias_out),
                                              _lib.current_stream()))


def head_compose_backward(G, tap_stride, c_stride, up_kernel, up_bias, head_w, S, taps, cin, cup, d_up_kernel, d_up_bias,
                          d_head_w):
    _lib.check(_lib.load().lisec_head_compose_backward(_lib.ptr(G), tap_stride, c_stride, _lib.ptr(up_kernel),
                                                       _lib.ptr(up_bias), _lib.ptr(head_w), _lib.ptr(S), taps, cin, cup,
                                                       _lib.ptr(d_up_kernel),
                                                       _lib.ptr(d_up_bias), _lib.ptr(d_head_w), _lib.current_stream()))


class HeadShuffle:
    """lisec_head_shuffle over fixed branch buffers: head += pixel-shuffled T_b (forward) / T_b = shuffled dhead (backward)."""

    def __init__(self, Ho, Wo, branches):
        self.Ho, self.Wo, self.n = Ho, Wo, len(branches)
        self.keep = [t for t, _ in branches]
        self.T = (ctypes.c_void_p * self.n)(*[t.data_ptr() for t, _ in branches])
        self.ps = (ctypes.c_int * self.n)(*[p for _, p in branches])

    def run(self, head, backward=False):
        _lib.check(_lib.load().lisec_head_shuffle(_lib.ptr(head), self.Ho, self.Wo, self.n, self.T, self.ps,
                                                  1 if backward else 0, _lib.current_stream()))
