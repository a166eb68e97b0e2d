"""The Lisec network (createModel, reference model_training.py:222-257) as a static schedule of
C-ABI calls on one HIP stream.

There is no graph tracer and no autograd: the layer order is fixed, so forward and backward are
explicit launch sequences over pre-allocated device buffers.  BatchNormalization(+ReLU) of a layer
is never materialised: every consumer applies scale/shift(+ReLU) while staging its input tile, so
only the raw convolution outputs ("y") live in HBM.  Per sample (batch 1, as model.fit(batch_size=1),
model_training.py:299):

    voxel sample -> VFE (sparse-exact) -> grid (D,H,W,64)
    mid_i : y_i = conv3d(u_{i-1}) ; stats ; u_i = relu(bn(y_i) @ Wd_i)              i = 1..3
    rpn_b : y_{b,j} = conv2d(relu(bn(y_{b,j-1}))) ; stats                           j = 0..q_b
    up_b  : concat[..., 256b:256(b+1)] = deconv(relu(bn(y_{b,q})))
    head  : (M,16) = concat @ [W_cls | W_reg] + bias     (cls = [:, :2], reg = [:, 2:])
"""
import os

import numpy as np
import torch

from . import _lib, ops
from .params import DECONVS, MID, RPN_BLOCKS, ParamStore, fold_depth
from .vfe import VFEStack


class ConvLayer:
    """One dense contraction: geometry + packed-weight slot + (optional) BatchNormalization."""

    def __init__(self, name, g, wname, pack, bias=None, bn=None, in_bn=None, in_relu=False, out_relu=False):
        self.name, self.g, self.wname, self.pack = name, g, wname, pack
        self.bias, self.bn, self.in_bn, self.in_relu, self.out_relu = bias, bn, in_bn, in_relu, out_relu
        self.M = g.Do * g.Ho * g.Wo
        self.nmb = ops.num_mblocks(g)


class LisecNet:
    def __init__(self, nx, ny, nz, maxPoints, params=None, device=None, compose_head=True):
        """compose_head: the three Conv2DTranspose branches and the 1x1 heads run as 16-channel contractions with composite
        kernels (csrc/head_fused.hip; exact by linearity, the (Ho,Wo,768) concat is never formed).  False keeps the
        layer-by-layer form (256-channel upsampling into the concat, then the 768 -> 16 heads) -- what the tests compare
        the collapsed form against."""
        self.device = device or _lib.require_gpu()
        self.compose_head = bool(compose_head)
        self.lib = _lib.load()
        if nx % 8 or ny % 8:
            raise ValueError("nx and ny must be multiples of 8 (three stride-2 RPN blocks)")
        self.H, self.W, self.D, self.T = nx, ny, nz, maxPoints
        self.dprime = fold_depth(nz)
        self.params = params if params is not None else ParamStore(self.device, dprime=self.dprime)
        if self.params.dprime != self.dprime:
            raise ValueError(f"variables are for a depth fold of {self.params.dprime}, nz={nz} folds to {self.dprime}")
        self.vfe = VFEStack(self.params, self.device)
        dev, f32 = self.device, torch.float32
        D, H, W = self.D, self.H, self.W
        self.act = {}          # name -> device tensor (raw conv outputs, mid outputs, grid, concat, head)
        self.bnstate = {}      # bn prefix -> float[4*C]
        self.layers = []

        def buf(name, *shape):
            self.act[name] = torch.empty(shape, dtype=f32, device=dev)
            return self.act[name]

        # the dense VFE output: only the dense form of the first Conv3D reads it (allocated on first use, 164 MB)
        self.grid_shape = (D, H, W, 64)
        # field form of the first Conv3D (csrc/field_conv.hip): sweeps with a voxel capacity (= min(points, cells)) up to
        # this never form the grid -- beyond ~40 % occupancy the dense contraction is the cheaper one
        # (LISEC_TUNING=field_conv=0 keeps the dense contraction for every sweep)
        self.field_conv = _lib.knob("field_conv", True)
        self.field_max_voxels = _lib.knob("field_max_voxels", 262144)
        self.field_ws = None
        self._used_field = False
        # ---- middle layers (model_training.py:236-238) ----------------------------------------
        d_in, prev = D, "grid"
        for i, (stride, pad) in enumerate(MID):
            d_out = (d_in + 2 * pad[0] - 3) // stride[0] + 1
            n = f"mid{i+1}"
            buf(n + ".y", d_out, H, W, 64)
            buf(n + ".u", d_out, H, W, 64)
            g = ops.geom(0, (d_in, H, W), (d_out, H, W), (3, 3, 3), stride, pad, 64, 64)
            self.layers.append(dict(kind="mid", name=n, src=prev, conv=ConvLayer(
                n + ".conv", g, n + ".conv.kernel", (27, 64, 64, 64 * 64, 64, 1), bias=n + ".conv.bias", bn=n + ".bn"),
                dense=ConvLayer(n + ".dense", ops.geom(0, (d_out, H, W), (d_out, H, W), (1, 1, 1), (1, 1, 1),
                                                       (0, 0, 0), 64, 64),
                                n + ".dense.kernel", (1, 64, 64, 0, 64, 1), in_bn=n + ".bn", out_relu=True)))
            d_in, prev = d_out, n + ".u"
        # Permute((2,3,4,1)) + Reshape (model_training.py:242-243): (D',H,W,64) -> (H,W,64*D'), channel c*D' + d.
        # D' = 1 (Constants.nz = 8): a view, nothing to do; otherwise one permuting copy ("fold") each way
        h, w, cin = H, W, 64 * d_in
        if d_in != 1:
            buf("fold", H, W, cin)
            self.fold_src, prev = prev, "fold"
        else:
            self.fold_src = None
        src, src_bn = prev, None
        Ho, Wo = H // 2, W // 2
        if not self.compose_head:
            buf("concat", Ho, Wo, 768)
        for b, (cout, q) in enumerate(RPN_BLOCKS):
            for j in range(q + 1):
                s = 2 if j == 0 else 1
                ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
                n = f"rpn{b+1}"
                buf(f"{n}.y{j}", ho, wo, cout)
                g = ops.geom(0, (1, h, w), (1, ho, wo), (1, 3, 3), (1, s, s), (0, 1, 1), cin, cout)
                self.layers.append(dict(kind="conv", name=f"{n}.conv{j}", src=src, dst=f"{n}.y{j}", conv=ConvLayer(
                    f"{n}.conv{j}", g, f"{n}.conv{j}.kernel", (9, cin, cout, cin * cout, cout, 1),
                    bias=f"{n}.conv{j}.bias", bn=f"{n}.bn{j}", in_bn=src_bn, in_relu=src_bn is not None)))
                src, src_bn, h, w, cin = f"{n}.y{j}", f"{n}.bn{j}", ho, wo, cout
            k, s = DECONVS[b]
            pad = (k - s) // 2
            if (h * s, w * s) != (Ho, Wo):
                raise ValueError("deconv output does not match the concat map")
            if k == s:
                # kernel == stride: no overlap -> a 1x1 GEMM whose columns (tap, n) are pixel-shuffled on store
                g = ops.geom(0, (1, h, w), (1, h, w), (1, 1, 1), (1, 1, 1), (0, 0, 0), cin, k * k * 256,
                             out_stride=768, ps=s, ps_channels=256)
                pack = (1, cin, k * k * 256, 0, 1, cin)
            else:
                g = ops.geom(1, (1, h, w), (1, h * s, w * s), (1, k, k), (1, s, s), (0, pad, pad), cin, 256,
                             out_stride=768)
                pack = (k * k, cin, 256, 256 * cin, 1, cin)
            L = dict(kind="deconv", name=f"up{b+1}", src=src, slot=b, k=k, s=s, pad=pad, hw=(h, w),
                     cin=cin, conv=ConvLayer(f"up{b+1}", g, f"up{b+1}.kernel", pack,
                                             bias=f"up{b+1}.bias", in_bn=src_bn, in_relu=True))
            if self.compose_head:
                # the branch and the heads as ONE contraction to 16 channels: Wc[tap][c][j] = sum_n W[tap][n][c] H[256b+n][j]
                taps = k * k
                L["Wc"] = torch.empty(taps * cin * 16, dtype=f32, device=dev)
                if k == s:
                    # kernel == stride: a 1x1 contraction with columns (tap, j), pixel-shuffled into the head afterwards
                    gf = ops.geom(0, (1, h, w), (1, h, w), (1, 1, 1), (1, 1, 1), (0, 0, 0), cin, taps * 16)
                    L["wc_strides"] = (16, taps * 16)                # Wc[c][tap*16 + j]
                    fpack = (1, cin, taps * 16, 0, taps * 16, 1)
                    L["T"] = torch.empty((h * w, taps * 16), dtype=f32, device=dev)
                else:
                    gf = ops.geom(1, (1, h, w), (1, h * s, w * s), (1, k, k), (1, s, s), (0, pad, pad), cin, 16)
                    L["wc_strides"] = (cin * 16, 16)                 # Wc[tap][c][j]
                    fpack = (taps, cin, 16, cin * 16, 16, 1)
                    L["T"] = None                                    # written straight into the head map
                L["conv"] = ConvLayer(f"up{b+1}.fused", gf, None, fpack, in_bn=src_bn, in_relu=True)
                L["up_kernel"], L["up_bias"] = f"up{b+1}.kernel", f"up{b+1}.bias"
            self.layers.append(L)
        buf("head", Ho, Wo, 16)
        self.Ho, self.Wo = Ho, Wo
        self.head_geom = ops.geom(0, (1, Ho, Wo), (1, Ho, Wo), (1, 1, 1), (1, 1, 1), (0, 0, 0), 768, 16)
        # ---- packed weights + BN state + stats scratch ---------------------------------------------
        self.packed = {}
        max_parts = 1
        for L in self.layers:
            for key in ("conv", "dense"):
                if key in L:
                    c = L[key]
                    self.packed[c.name] = torch.empty(ops.packed_floats(c.pack[0], c.pack[1], c.pack[2]),
                                                      dtype=f32, device=dev)
                    if c.bn:
                        self.bnstate[c.bn] = torch.zeros(4 * c.g.Cout, dtype=f32, device=dev)
                        max_parts = max(max_parts, c.nmb * 2 * c.g.Cout)
        self.packed["head"] = torch.empty(ops.packed_floats(1, 768, 16), dtype=f32, device=dev)
        # Winograd F(2x2, 3x3) form (csrc/wino.hip) of the stride-1 3x3 contractions that fill the chip in it (>= 128 blocks of
        # 8 x 8 tiles x 64 channels: the Conv3D blocks behind the first and the stride-1 Conv2Ds of RPN block 1): 4 / 9 of the
        # multiplications.  LISEC_TUNING winograd: bit 0 = forward calls, bit 1 = data gradients of the Conv2Ds, bit 2 = data gradients of the Conv3D
        # blocks (their Dense(64) gradient then runs as a launch of its own), bit 3 = weight gradients of the Conv3D blocks
        # (csrc/wino_wgrad.hip); 0 keeps the direct kernels
        self.winograd = _lib.knob("winograd", 15)
        self.packed_wu, self.packed_wu_t = {}, {}
        for L in self.layers:
            c = L["conv"]
            g = c.g
            if L["kind"] == "deconv" or L["src"] == "grid" or not self.winograd:
                continue
            blocks = g.Do * ((g.Ho + 15) // 16) * ((g.Wo + 15) // 16) * ((g.Cout + 63) // 64)
            if blocks >= 128 and ops.winograd_supported(g, in_bn=c.in_bn is not None, flags=ops.IN_RELU if c.in_relu else 0):
                if self.winograd & 1:
                    self.packed_wu[c.name] = torch.empty(ops.winograd_packed_floats(g.KD, g.Cin, g.Cout), dtype=f32, device=dev)
                # data gradient: kept on the direct kernel where the Dense(64) gradient of the block below rides on its tile
                # (lisec_conv_extras.tail_w: the middle blocks) -- measured equal there, and the tail would cost a launch
                if (self.winograd & 2 and L["kind"] == "conv") or (self.winograd & 4 and L["kind"] == "mid"):
                    self.packed_wu_t[c.name] = torch.empty(ops.winograd_packed_floats(g.KD, g.Cout, g.Cin), dtype=f32,
                                                           device=dev)
        self.head_w = torch.empty(768, 16, dtype=f32, device=dev)
        self.head_b = torch.empty(16, dtype=f32, device=dev)
        self.fused_bias = torch.empty(16, dtype=f32, device=dev)     # b' = head bias + the branch biases through H
        if self.compose_head:
            self._shuffle = ops.HeadShuffle(Ho, Wo, [(L["T"], L["s"]) for L in self.layers
                                                     if L["kind"] == "deconv" and L["T"] is not None])
        self.parts = torch.empty(max_parts, dtype=torch.float64, device=dev)
        self._sinks, self._bsinks = {}, {}
        self.early_pack = _lib.knob("pack_early", True)
        # The second HIP stream.  Backward: weight gradients are leaves of the graph and run beside the BN-backward /
        # data-gradient chain (the RPN layers are too small to fill 256 CUs on their own).  Forward and backward: the
        # Conv2DTranspose branches of RPN blocks 1 and 2 (model_training.py:246,249) only meet the rest of the network at the
        # concat, so they run beside the next block's small convolutions instead of in front of them.
        # ROCm multiplexes same-priority streams onto a few hardware queues round-robin, so a plain second stream
        # can land on the main stream's queue (it does once RCCL has made its own streams) and then nothing
        # overlaps; a different priority level always gets its own hardware queue.
        # (Round 4: with GPU_MAX_HW_QUEUES=8 -- set when lisec_amd is imported -- a stream of the SAME priority gets a queue
        # of its own, and that is the faster arrangement: the high-priority queue was served first whenever it held a ready
        # packet, which starved the chain during the head phase; one rank through RCCL 5.13 -> 4.25 ms, one GPU +0.8 %.)
        self.side = torch.cuda.Stream(device=dev, priority=_lib.knob("side_priority", 0))

        self.branch_overlap = _lib.knob("branch_overlap", True)
        self._tail_ok = {}
        self._fold_ok = {}
        # BatchNormalization backward applied on load by the next data gradient: 0 = never (default: measured equal at best),
        # 1 = on the w-halo kernels of the small maps (see _fold_supported), 2 = everywhere (4 % slower)
        self.fold_bn_bwd = _lib.knob("fold_bn_bwd", 0)
        self.early_sgd = _lib.knob("early_sgd", True)           # RPN + head variables updated under the rest of the backward pass
        self._early_from = None
        self.dense_wgrad_late = _lib.knob("dense_wgrad_late", True)   # Dense weight gradient behind the block's ring weight gradient
        self.mid_wgrad_first = _lib.knob("mid_wgrad_first", True)   # ring weight gradient enqueued before the block's data gradient
        self.pack_mid_first = _lib.knob("pack_mid_first", True)   # the forward waits for the middle blocks' repack only
        self.dense_dw = _lib.knob("dense_dw", True)
        self.dense_dw_late = _lib.knob("dense_dw_late", True)   # their slab sums behind the last weight gradient of the second stream      # Dense(64) weight gradients ride on the Dense data gradients
        self.dense_dw_slabs = {}                         # middle block -> slabs of lisec_conv_extras.dense_dw
        self.fuse_dense_bwd = _lib.knob("fuse_dense_bwd", True)   # Dense(64) data gradients ride on the tile of the block above
        self.chain_first = _lib.knob("chain_first", True)      # head phase: the chain's contraction is enqueued before the leaves
        self._fwd_events = {}
        self._packed_version = -1
        self.params_version = 0
        self.state_version = 0
        self._folded = {}
        self._train_ready = False
        # optimizer iteration count: the device copy drives the learning-rate decay (lisec_sgd_nesterov_step_dev), so
        # that a captured step can be replayed; the host mirror is what save()/load_model() and the tests read
        self._iter_dev = torch.zeros(2, dtype=torch.int64, device=dev)
        self._iterations = 0
        self.loss_out = torch.zeros(3, dtype=f32, device=dev)

    @property
    def iterations(self):
        return self._iterations

    @iterations.setter
    def iterations(self, k):
        self._iterations = int(k)
        self._iter_dev.copy_(torch.tensor([int(k), 0], dtype=torch.int64))

    # ------------------------------------------------------------------------------------------------
    def _pack_all(self, after_main=None):
        """Repack theta into the kernels' [tap][K/4][N][4] layout (after every optimizer step)."""
        if self._packed_version == (self.params_version, self.params.version):
            if after_main is not None:
                after_main()
            return
        p = self.params
        if getattr(self, "_head_merge", None) is None:     # Keras-shaped head variables -> the merged (768,16) layout
            self._head_merge = ops.CopyTable([(p.view("cls.kernel")[0, 0], self.head_w[:, :2]),
                                              (p.view("reg.kernel")[0, 0], self.head_w[:, 2:]),
                                              (p.view("cls.bias"), self.head_b[:2]),
                                              (p.view("reg.bias"), self.head_b[2:])], self.device)
        self._head_merge.run()
        if getattr(self, "_pack_table", None) is None:
            entries, rest, late = [], [], []
            for L in self.layers:
                for key in ("conv", "dense"):
                    if key in L:
                        c = L[key]
                        (late if not c.wname else entries if L["kind"] == "mid" else rest).append(
                            ((p.view(c.wname) if c.wname else L["Wc"]), self.packed[c.name]) + tuple(c.pack))
            self._pack_table = ops.PackTable(entries, self.device)
            self._pack_table_rest = ops.PackTable(rest, self.device)
            self._pack_table_wc = ops.PackTable(late, self.device) if late else None

        def wino_packs(mid):
            for L in self.layers:
                c = L["conv"]
                if c.name in self.packed_wu and (L["kind"] == "mid") == mid:
                    g = c.g
                    ops.pack_weights_winograd(p.view(c.wname), g.KD, g.Cin, g.Cout, g.Cin * g.Cout, g.Cout, 1,
                                              out=self.packed_wu[c.name])
        # the kernels of the middle blocks first: the first contraction of the forward pass waits for these only (after_main
        # records that point when the repack runs early on the second stream: 6 small kernels, done before the VFE is -- with
        # every variable kernel in front of that point the main stream stood 22 us per step).  The RPN's kernels, the composite
        # kernels of the collapsed heads -- three compose launches in front of their pack -- and the transposed set follow;
        # the first Conv2D of the RPN, ~0.5 ms into the step, waits for all of them (_pack_late)
        self._pack_table.run()
        wino_packs(True)
        if after_main is not None and self.pack_mid_first:
            after_main()
        self._pack_table_rest.run()
        wino_packs(False)
        if after_main is not None and not self.pack_mid_first:
            after_main()
        if self.compose_head:
            self._compose_all()
        if self._pack_table_wc is not None:
            self._pack_table_wc.run()
        if not self.compose_head:
            ops.pack_weights(self.head_w, 1, 768, 16, 0, 16, 1, out=self.packed["head"])
        self._packed_version = (self.params_version, self.params.version)

    def _compose_all(self):
        """Composite kernels of the three upsampling branches with the heads, and the composite bias (head_fused.hip)."""
        p = self.params
        bias_in = self.head_b
        for L in self.layers:
            if L["kind"] != "deconv":
                continue
            b, ts, cs = L["slot"], *L["wc_strides"]
            ops.head_compose(p.view(L["up_kernel"]), p.view(L["up_bias"]), self.head_w[256 * b:256 * (b + 1)], L["k"] * L["k"],
                             L["cin"], 256, L["Wc"], ts, cs, bias_in=bias_in, bias_out=self.fused_bias)
            bias_in = self.fused_bias

    def _fwd_sink(self, c):
        """BnSink of a conv layer's BatchNormalization: batch statistics summed and finalised inside the conv call."""
        s = self._sinks.get(c.bn)
        if s is None:
            p = self.params
            s = self._sinks[c.bn] = ops.BnSink(c.g.Cout, c.M, self.device, gamma=p.view(c.bn + ".gamma"),
                                               beta=p.view(c.bn + ".beta"), moving_mean=p.view(c.bn + ".moving_mean"),
                                               moving_var=p.view(c.bn + ".moving_variance"), unbiased=True,
                                               bnstate=self.bnstate[c.bn])
        return s

    def _bn_after(self, c, training):
        p = self.params
        C = c.g.Cout
        if training:
            # the statistics were finalised by the last workgroup of the conv call (lisec_bn_sink): nothing to launch
            self.state_version += 1               # moving statistics moved, bnstate holds batch statistics
        elif self._folded.get(c.bn) != (self.params_version, self.state_version, p.version):
            # inference: scale/shift from the moving statistics, folded once per weight version (not per sweep)
            ops.bn_fold(p.view(c.bn + ".gamma"), p.view(c.bn + ".beta"), p.view(c.bn + ".moving_mean"),
                        p.view(c.bn + ".moving_variance"), C, self.bnstate[c.bn])
            self._folded[c.bn] = (self.params_version, self.state_version, p.version)

    def _run_conv(self, c, x, out, training, ws_tag="main"):
        p = self.params
        flags = (ops.IN_RELU if c.in_relu else 0) | (ops.OUT_RELU if c.out_relu else 0)
        sink = self._fwd_sink(c) if (c.bn and training) else None
        if c.name in self.packed_wu:
            ops.conv_forward_winograd(c.g, x, self.packed_wu[c.name], out, bias=p.view(c.bias) if c.bias else None,
                                      in_bn=self.bnstate[c.in_bn] if c.in_bn else None, flags=flags, sink=sink)
        else:
            ops.conv_forward(c.g, x, self.packed[c.name], out, bias=p.view(c.bias) if c.bias else None,
                             in_bn=self.bnstate[c.in_bn] if c.in_bn else None, flags=flags, sink=sink, ws_tag=ws_tag)
        if c.bn:
            self._bn_after(c, training)

    def _event(self, name):
        ev = self._fwd_events.get(name)
        if ev is None:
            ev = self._fwd_events[name] = self._new_event()
        return ev

    def _new_event(self):
        """Fork / join events: device-scope release, recorded and waited for through the library (lisec_event_record),
        so that a step plan sees them."""
        return _lib.DeviceEvent()

    @staticmethod
    def _record(ev, stream):
        ev.record(stream.cuda_stream)

    @staticmethod
    def _wait(ev, stream):
        ev.wait(stream.cuda_stream)

    def _mark(self, name):
        """Diagnostic (tools/phase_times.py): with self.phase_marks a dict, a timing event is recorded on the main stream
        here -- through the library, so that a step plan replays it -- and in-step phase durations can be read back
        WITHOUT a profiler (rocprofv3's kernel trace makes the host the bottleneck wherever many small kernels are launched
        and shows gaps that a plain run does not have)."""
        marks = getattr(self, "phase_marks", None)
        if marks is None:
            return
        ev = marks.get(name)
        if ev is None:
            ev = marks[name] = _lib.DeviceEvent(timing=True)
        ev.record(_lib.current_stream())

    def dense_grid(self, rewrite=True):
        """The dense (D,H,W,64) VFE output; rewrite: fill it from the last forward's per-voxel values (the field form
        of the first Conv3D never writes it)."""
        if "grid" not in self.act:
            self.act["grid"] = torch.empty(self.grid_shape, dtype=torch.float32, device=self.device)
        if rewrite:
            self.vfe.rewrite_grid(self.act["grid"])
        return self.act["grid"]

    def forward(self, sample, training=False):
        """sample: VoxelSample of one lidar sweep.  Returns (cls (1,Ho,Wo,2), reg (1,Ho,Wo,14)) device views."""
        if sample.grid_shape != (self.D, self.H, self.W) or sample.cfg.sampleSize != self.T:
            raise ValueError("voxel sample does not match the model's grid")
        prev_pin = _lib.pin_stream(torch.cuda.current_stream().cuda_stream)   # one stream query for the whole schedule
        try:
            return self._forward(sample, training)
        finally:
            _lib.pin_stream(prev_pin)

    def _forward(self, sample, training):
        self._mark("step:begin")
        pending = getattr(self, "_pack_pending", False)
        if pending and (self._packed_version != (self.params_version, self.params.version)):
            self._wait(self._pack_late, torch.cuda.current_stream())     # variables changed since the early repack
            self._pack_pending = pending = False
            self._late_pending = False
        if not pending:
            self._pack_all()
        a = self.act
        use_field = self._used_field = self.field_conv and sample.cap <= self.field_max_voxels
        if use_field:
            first = self.layers[0]["conv"]
            need = ops.conv_field_forward_workspace_bytes(first.g, sample.cap)
            if self.field_ws is None or self.field_ws.numel() < need:
                self.field_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.bump_alloc_generation()       # recorded step plans hold the old address
            self.vfe.forward(sample, training, dense=False)
        else:
            self.vfe.forward(sample, training, out=self.dense_grid(rewrite=False))
        self._mark("step:vfe done")
        if pending:
            # the repack of this step's weights was enqueued on the second stream right after the last optimizer step
            # and ran under this sweep's voxeliser and VFE; the first contraction is the first reader
            self._wait(self._pack_done, torch.cuda.current_stream())
            self._pack_pending = False
        side_used = False
        self._mark("fwd:start")
        for L in self.layers:
            self._mark("fwd:before " + L["name"])
            if L["kind"] == "mid":
                n = L["name"]
                if L["src"] == "grid" and use_field:
                    c = L["conv"]
                    ops.conv_field_forward(c.g, self.vfe.saved_field("vout"), self.vfe.saved_field("delta"), sample,
                                           self.packed[c.name], a[n + ".y"], self.field_ws, bias=self.params.view(c.bias),
                                           sink=self._fwd_sink(c) if training else None)
                    self._bn_after(c, training)
                else:
                    self._run_conv(L["conv"], a[L["src"]], a[n + ".y"], training)
                self._run_conv(L["dense"], a[n + ".y"], a[n + ".u"], training)
            elif L["kind"] == "conv":
                if getattr(self, "_late_pending", False):
                    # first reader of a kernel repacked behind the middle blocks' (see _pack_all)
                    self._wait(self._pack_late, torch.cuda.current_stream())
                    self._late_pending = False
                if L["src"] == "fold":
                    ops.fold_depth(a[self.fold_src], a["fold"], self.dprime, self.H * self.W, 64)
                self._run_conv(L["conv"], a[L["src"]], a[L["dst"]], training)
            else:
                b = L["slot"]
                if self.compose_head:
                    # 16-channel contraction: straight into the head map (with the composite bias) or into the branch's
                    # (tap, j) columns, added to the head by the shuffle pass below
                    c = L["conv"]
                    dst = a["head"] if L["T"] is None else L["T"]

                    def run(ws_tag, c=c, L=L, dst=dst):
                        ops.conv_forward(c.g, a[L["src"]], self.packed[c.name], dst,
                                         bias=self.fused_bias if L["T"] is None else None, in_bn=self.bnstate[c.in_bn],
                                         flags=ops.IN_RELU, ws_tag=ws_tag)
                else:
                    def run(ws_tag, L=L, b=b):
                        self._run_conv(L["conv"], a[L["src"]], a["concat"][:, :, 256 * b:], training, ws_tag=ws_tag)
                if getattr(self, "_late_pending", False) and not (self.branch_overlap and b < len(DECONVS) - 1):
                    # first reader on THIS stream of a kernel repacked late on the second one (composite kernels, and the
                    # transposed set the backward reads)
                    self._wait(self._pack_late, torch.cuda.current_stream())
                    self._late_pending = False
                if self.branch_overlap and b < len(DECONVS) - 1:
                    # an upsampling branch that is not the last: beside the next block, on the second stream
                    main = torch.cuda.current_stream()
                    fork = self._event("fwd_fork%d" % b)
                    self._record(fork, main)
                    self._wait(fork, self.side)
                    pin = _lib.pin_stream(self.side.cuda_stream)
                    try:
                        run("side")
                    finally:
                        _lib.pin_stream(pin)
                    side_used = True
                else:
                    run("main")
        if getattr(self, "_late_pending", False):
            self._wait(self._pack_late, torch.cuda.current_stream())
            self._late_pending = False
        if side_used:
            join = self._event("fwd_join")
            self._record(join, self.side)
            self._wait(join, torch.cuda.current_stream())
        if self.compose_head:
            self._shuffle.run(a["head"])
        else:
            ops.conv_forward(self.head_geom, a["concat"], self.packed["head"], a["head"], bias=self.head_b)
        head = a["head"]
        return head[None, :, :, :2], head[None, :, :, 2:]

    # ------------------------------------------------------------------------------------------------
    # training: explicit backward schedule (what Keras' fit() derives by autograd, model_training.py:299)
    def _prepare_training(self):
        if self._train_ready:
            return
        dev, f32 = self.device, torch.float32
        p = self.params
        self.grad = torch.zeros_like(p.theta)
        self.velocity = torch.zeros_like(p.theta)
        self.dact = {}
        for name, t in self.act.items():
            if name.endswith(".u") or name in ("concat", "head", "fold") or ".y" in name:
                self.dact[name] = torch.empty_like(t)
        self.packed_t = {}
        self.dgeom = {}
        ws_bytes = ops.wgrad_workspace_bytes(self.head_geom)
        Ho, Wo = self.Ho, self.Wo
        for L in self.layers:
            c = L["conv"]
            g = c.g
            ntaps = g.KD * g.KH * g.KW
            ws_bytes = max(ws_bytes, ops.wgrad_workspace_bytes(g))
            if L["kind"] == "deconv" and self.compose_head:
                # the 16-channel contraction's gradients: G = dL/dWc by the ordinary weight gradient, the data gradient with
                # the composite kernel transposed; dy is the head gradient itself or its (tap, j) columns (dT)
                k, sd, pad, (h, w), cin = L["k"], L["s"], L["pad"], L["hw"], L["cin"]
                taps = k * k
                L["G"] = torch.empty(taps * cin * 16, dtype=f32, device=dev)
                if k == sd:
                    L["dT"] = torch.empty_like(L["T"])
                    self.dgeom[c.name] = ops.geom(0, (1, h, w), (1, h, w), (1, 1, 1), (1, 1, 1), (0, 0, 0), taps * 16, cin)
                    spec = (1, taps * 16, cin, 0, 1, taps * 16)
                else:
                    L["dT"] = None
                    self.dgeom[c.name] = ops.geom(0, (1, Ho, Wo), (1, h, w), (1, k, k), (1, sd, sd), (0, pad, pad), 16, cin)
                    spec = (taps, 16, cin, cin * 16, 1, 16)
                self.packed_t[c.name] = (torch.empty(ops.packed_floats(spec[0], spec[1], spec[2]), dtype=f32, device=dev),
                                         spec)
            elif L["kind"] == "deconv":
                # data gradient of a transposed conv = plain strided conv over dY (K = out, N = in)
                k, sd, pad, (h, w), cin = L["k"], L["s"], L["pad"], L["hw"], L["cin"]
                ntaps = k * k
                self.dgeom[c.name] = ops.geom(0, (1, Ho, Wo), (1, h, w), (1, k, k), (1, sd, sd), (0, pad, pad),
                                              256, cin, in_stride=768)
                self.packed_t[c.name] = (torch.empty(ops.packed_floats(ntaps, 256, cin), dtype=f32, device=dev),
                                         (ntaps, 256, cin, 256 * cin, cin, 1))
                if k == sd:
                    # weight gradient with swapped roles: gather dY (stride s), contract against the input rows
                    L["wgeom"] = self.dgeom[c.name]
                    ws_bytes = max(ws_bytes, ops.wgrad_workspace_bytes(L["wgeom"]))
            else:
                self.dgeom[c.name] = ops.geom(1, (g.Do, g.Ho, g.Wo), (g.Di, g.Hi, g.Wi), (g.KD, g.KH, g.KW),
                                              (g.sd, g.sh, g.sw), (g.pd, g.ph, g.pw), g.Cout, g.Cin)
                self.packed_t[c.name] = (torch.empty(ops.packed_floats(ntaps, g.Cout, g.Cin), dtype=f32, device=dev),
                                         (ntaps, g.Cout, g.Cin, g.Cin * g.Cout, 1, g.Cout))
            if "dense" in L:
                d = L["dense"]
                ws_bytes = max(ws_bytes, ops.wgrad_workspace_bytes(d.g))
                self.dgeom[d.name] = d.g          # 1x1: the data gradient is the same geometry with W^T
                self.packed_t[d.name] = (torch.empty(ops.packed_floats(1, 64, 64), dtype=f32, device=dev),
                                         (1, 64, 64, 0, 1, 64))
                self.dact[L["name"] + ".z"] = torch.empty_like(self.act[L["name"] + ".y"])
        # first middle layer: exact sparse backward (csrc/sparse_grid.hip) -- never forms the 164 MB grid gradient
        first = self.layers[0]["conv"]
        self.mid1_S = torch.empty(27 * 64, dtype=f32, device=dev)
        self.g_all = torch.empty(64, dtype=f32, device=dev)
        self.tapsum_ws = torch.empty(ops.tap_sums_workspace_bytes(first.g), dtype=torch.uint8, device=dev)
        self.dout_rows = None
        self.rows_queue = torch.zeros(2, dtype=torch.int32, device=dev)      # tile counter of the row-list data gradient
        self.head_dgeom = ops.geom(0, (1, Ho, Wo), (1, Ho, Wo), (1, 1, 1), (1, 1, 1), (0, 0, 0), 16, 768)
        if not self.compose_head:
            self.packed_t["head"] = (torch.empty(ops.packed_floats(1, 16, 768), dtype=f32, device=dev), None)
        else:
            self._dshuffle = ops.HeadShuffle(Ho, Wo, [(L["dT"], L["s"]) for L in self.layers
                                                      if L["kind"] == "deconv" and L["dT"] is not None])
        self.head_dw = torch.empty(768, 16, dtype=f32, device=dev)
        self.up_db = torch.empty(768, dtype=f32, device=dev)
        self._fork_events, self._join_event = [], None
        # conv outputs that sit under a BatchNormalization(+ReLU), and how many layers read each of them
        self.bn_of, self.consumers = {}, {}
        nparts = 1
        for L in self.layers:
            self.consumers[L["src"]] = self.consumers.get(L["src"], 0) + 1
            if L["kind"] == "conv":
                self.bn_of[L["dst"]] = (L["conv"].bn, L["conv"].g.Cout)
        for L in self.layers:
            if L["src"] in self.bn_of:
                nparts = max(nparts, ops.num_mblocks_bwd(self.dgeom[L["conv"].name]) * 2 * self.bn_of[L["src"]][1])
            if "dense" in L:
                nparts = max(nparts, ops.num_mblocks_bwd(self.dgeom[L["dense"].name]) * 2 * 64)
        self.bparts = torch.empty(nparts, dtype=torch.float64, device=dev)
        self.head_db = torch.empty(16, dtype=f32, device=dev)
        # the stride-1 convolutions of an RPN block (model_training.py:210-214) share ONE weight-gradient launch: maps of
        # 1 250 - 20 000 positions fill a fraction of the chip each, and as leaves of the backward pass they can wait for each
        # other (lisec_conv_wgrad_batched)
        # BatchNormalization backward folded into the NEXT data gradient's load (fold_bn_bwd): the chain reads the raw
        # gradient d[dst] and applies the backward on load; the weight gradient (second stream) reads the applied gradient
        # from a buffer of its own, written by an apply launch on THAT stream -- off the chain
        self.dyb = {}
        if self.fold_bn_bwd:
            for L in self.layers:
                if L["kind"] == "conv":
                    self.dyb[L["dst"]] = torch.empty_like(self.dact[L["dst"]])
        self.wgrad_batches = {}
        if _lib.knob("wgrad_batch", True):
            for b in range(len(RPN_BLOCKS)):
                convs = [L for L in self.layers if L["kind"] == "conv" and L["name"].startswith(f"rpn{b+1}.conv")
                         and L["name"] != f"rpn{b+1}.conv0"]
                items = [(L["conv"].g, self.act[L["src"]], self.dyb.get(L["dst"], self.dact[L["dst"]]),
                          p.grad_view(self.grad, L["conv"].wname), self.bnstate[L["conv"].in_bn], ops.IN_RELU, False)
                         for L in convs]
                if 2 <= len(items) <= 6:
                    batch = ops.WgradBatch(items)
                    ws_bytes = max(ws_bytes, batch.workspace_bytes())
                    self.wgrad_batches[convs[0]["name"]] = (batch, {L["name"] for L in convs})
        # zero-filled: the head of the workspace holds the arrival counters of the slab-combining kernels
        self.wgrad_ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
        # weight gradients of the Conv3D blocks behind the first in the Winograd form (winograd bit 3): one slab workspace for
        # all of them (they run one after the other on the second stream)
        self.wino_wgrad_ws = {}
        if self.winograd & 8:
            mids = [L["conv"] for L in self.layers if L["kind"] == "mid" and L["src"] != "grid"
                    and ops.wgrad_winograd_supported(L["conv"].g)]
            if mids:
                shared = torch.empty(max(ops.wgrad_winograd_workspace_bytes(c.g) for c in mids), dtype=torch.uint8, device=dev)
                self.wino_wgrad_ws = {c.name: shared for c in mids}
        # Dense(64) weight gradients carried by the Dense data gradients (dense_dw): one slab buffer per middle block -- the sum
        # runs on the second stream, possibly after the next block's data gradient has started writing its own
        self.dense_dw_slabs = {}
        if self.dense_dw:
            nslabs = ops.dense_dw_slabs()
            for L in self.layers:
                if L["kind"] == "mid" and (L["conv"].M + 127) // 128 >= nslabs:
                    self.dense_dw_slabs[L["name"]] = torch.empty(nslabs * 4096, dtype=torch.float32, device=dev)
        self._packed_t_version = -1
        self._train_ready = True

    def _bwd_sink(self, bn_name, C, n_rows):
        """BnSink of the backward of one BatchNormalization: (sum dz, sum dz*yhat) -> dgamma, dbeta, coefficients."""
        s = self._bsinks.get(bn_name)
        if s is None:
            p = self.params
            s = self._bsinks[bn_name] = ops.BnSink(C, n_rows, self.device, dgamma=p.grad_view(self.grad, bn_name + ".gamma"),
                                                   dbeta=p.grad_view(self.grad, bn_name + ".beta"))
        return s

    def _pack_all_t(self):
        if self._packed_t_version == (self.params_version, self.params.version):
            return
        p = self.params
        if self.compose_head and self._packed_version != (self.params_version, self.params.version):
            self._pack_all()                 # the composite kernels (Wc) are made there
        if getattr(self, "_pack_table_t", None) is None:
            entries = []
            for L in self.layers:
                for key in ("conv", "dense"):
                    if key in L:
                        c = L[key]
                        buf, spec = self.packed_t[c.name]
                        entries.append((p.view(c.wname) if c.wname else L["Wc"], buf) + tuple(spec))
            if not self.compose_head:
                entries.append((self.head_w, self.packed_t["head"][0], 1, 16, 768, 0, 1, 16))
            self._pack_table_t = ops.PackTable(entries, self.device)
        self._pack_table_t.run()
        for L in self.layers:
            c = L["conv"]
            if c.name in self.packed_wu_t:
                g = c.g                          # K = forward Cout, N = forward Cin, taps mirrored
                ops.pack_weights_winograd(p.view(c.wname), g.KD, g.Cout, g.Cin, g.Cin * g.Cout, 1, g.Cout, flip=True,
                                          out=self.packed_wu_t[c.name])
        self._packed_t_version = (self.params_version, self.params.version)

    def backward(self, y_cls, y_reg, loss="mse", grad_scale=1.0, rpn_grads_ready=None, side_filler=None):
        """y_cls (Ho,Wo,2), y_reg (Ho,Wo,14): float32 device tensors.  Fills self.grad (layout of theta)
        and self.loss_out = [total, class, regression].  Must follow forward(training=True).
        side_filler: optional callable issued on the second stream behind the head-phase leaves, where that stream has
        nothing to do for ~200 us (the weight gradients of the last RPN block wait for its chain): independent work such as
        the NEXT sweep's voxelisation (PipelinedStep).
        rpn_grads_ready(lo, hi): optional hook, called (inside the second stream's context) as soon as the
        gradients of every RPN/head variable -- theta[lo:hi], 94 % of the parameters -- are final, while the
        middle layers and the VFE are still being differentiated: data parallelism starts its all-reduce there."""
        prev_pin = _lib.pin_stream(torch.cuda.current_stream().cuda_stream)
        try:
            return self._backward(y_cls, y_reg, loss, grad_scale, rpn_grads_ready, side_filler)
        finally:
            _lib.pin_stream(prev_pin)

    def _fold_supported(self, c, dst):
        """Does the data gradient of conv `c` take the BatchNormalization backward of its input gradient on load?  (asked of
        the library once per layer)"""
        ok = self._fold_ok.get(c.name)
        if ok is None:
            try:
                sink = self._bwd_sink(c.bn, c.g.Cout, c.M)
                plan = ops.conv_plan(self.dgeom[c.name], fold=(self.act[dst], self.bnstate[c.bn], sink.coef, True))
                # measured per layer (tools/phase_times.py, round 4): the fold pays where the A tile is staged once per three
                # K steps (w-halo kernels of the small maps: -3.5 us per layer); on the generic gather (a second operand
                # load in EVERY step of a lone wave) and on the 3-per-CU kernels it costs 4 - 44 us per layer
                ok = self.fold_bn_bwd == 2 or (plan["kernel"] == "halo3" and plan["double_buffered"] == 1)
            except _lib.LisecError:
                ok = False
            self._fold_ok[c.name] = ok
        return ok

    def _tail_supported(self, c, dst_name, winograd=False):
        """Can the Dense data gradient of block dst_name[:-2] ride on the data gradient of conv `c`?  (asked of the library
        once per layer: lisec_conv_plan_query refuses geometries the two-line w-halo kernel does not serve;
        lisec_conv_winograd_supported answers for the Winograd form)"""
        ok = self._tail_ok.get((c.name, winograd))
        if ok is None and winograd:
            # (LISEC_TUNING winograd_tail: the tail on the Winograd epilogue is built and parity-tested
            # (tests/test_gpu_winograd.py) and measured 0.5 % SLOWER in the step than the separate HBM-bound Dense launch,
            # which hides beside the MFMA-bound weight gradients of the second stream: off)
            n = dst_name[:-2]
            Ln = {L["name"]: L for L in self.layers}.get(n)
            ok = False
            if Ln is not None and "dense" in Ln and self.dgeom[c.name].Cout == 64 and _lib.knob("winograd_tail", False):
                cn, dn = Ln["conv"], Ln["dense"]
                ok = ops.winograd_supported(self.dgeom[c.name], out_mask=self.act[dst_name],
                                            bwd=(self.act[n + ".y"], self.bnstate[cn.bn], False),
                                            sink=self._bwd_sink(cn.bn, 64, cn.M),
                                            tail=(self.packed_t[dn.name][0], self.dact[n + ".z"]))
            self._tail_ok[(c.name, winograd)] = ok
        if ok is None:
            n = dst_name[:-2]
            Ln = {L["name"]: L for L in self.layers}.get(n)
            ok = False
            if Ln is not None and "dense" in Ln and self.dgeom[c.name].Cout == 64:
                cn, dn = Ln["conv"], Ln["dense"]
                try:
                    ops.conv_plan(self.dgeom[c.name], out_mask=self.act[dst_name],
                                  bwd=(self.act[n + ".y"], self.bnstate[cn.bn], False), sink=self._bwd_sink(cn.bn, 64, cn.M),
                                  tail=(self.packed_t[dn.name][0], self.dact[n + ".z"]))
                    ok = True
                except _lib.LisecError:
                    ok = False
            self._tail_ok[(c.name, winograd)] = ok
        return ok

    def _backward(self, y_cls, y_reg, loss, grad_scale, rpn_grads_ready, side_filler=None):
        self._prepare_training()
        self._pack_all_t()
        p, a, d, G = self.params, self.act, self.dact, self.grad
        M = self.Ho * self.Wo
        sample = self.vfe._sample
        first = self.layers[0]["conv"]
        rcap = max(sample.cap, 1)                 # an empty sweep still needs a non-zero row-list capacity
        need = ops.wgrad_workspace_bytes(self.dgeom[first.name], rcap)
        if need > self.wgrad_ws.numel() or self.dout_rows is None or self.dout_rows.shape[0] < sample.cap + 1:
            torch.cuda.synchronize()               # (re)size scratch that depends on the cloud's capacity
            if need > self.wgrad_ws.numel():
                self.wgrad_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self.dout_rows = torch.empty((sample.cap + 1, 64), dtype=torch.float32, device=self.device)
            _lib.bump_alloc_generation()           # recorded step plans hold the old addresses
        main = torch.cuda.current_stream()

        side_handle = self.side.cuda_stream
        events = self._fork_events
        nfork = [0]

        pending = []

        marked = []

        def mark_fork():
            """Records the fork event NOW; the next flush_side() waits for this one instead of recording its own.  Lets the
            chain's next kernel be ENQUEUED before the second stream's work although that work does not depend on it."""
            if nfork[0] == len(events):
                events.append(self._new_event())
            ev = events[nfork[0]]                       # events are reused step after step
            nfork[0] += 1
            self._record(ev, main)
            marked.append(ev)

        def flush_side():
            """Records ONE event on the main stream and runs every pending closure on the second stream behind it."""
            if not pending:
                return
            if not marked:
                mark_fork()
            ev = marked.pop()
            self._wait(ev, self.side)
            pin = _lib.pin_stream(side_handle)
            try:
                for fn, torch_ops in pending:
                    if torch_ops:
                        with torch.cuda.stream(self.side):
                            fn()
                    else:
                        fn()
            finally:
                _lib.pin_stream(pin)
                del pending[:]

        skip_leaves = _lib.knob("skip_leaves", False)      # measurement aid (WRONG gradients): the chain alone

        def on_side(fn, torch_ops=False):
            """Runs fn's launches on the second stream after everything issued so far on the main one.  C-ABI launches
            take the pinned handle; only a fn that also issues torch / torch.distributed work needs torch's (slow)
            stream context."""
            if skip_leaves and not torch_ops:
                flush_side()
                return
            pending.append((fn, torch_ops))
            flush_side()

        self._mark("bwd:start")
        kind = {"mse": 0, "smoothl1_ce": 1}[loss]
        ops.rpn_loss(a["head"], y_cls, y_reg, M, kind, d["head"], self.loss_out, grad_scale=grad_scale)
        # ---- heads (model_training.py:254-255) ---------------------------------------------------
        # only the data gradient is on the way to the rest of the backward pass: the heads' weight and bias gradients and
        # the deconv bias gradients (column sums of the concat gradient) are leaves and go to the second stream
        if getattr(self, "_head_split", None) is None:     # merged head gradients -> the Keras-shaped slots of G
            self._head_split = ops.CopyTable([(self.head_dw[:, :2], p.grad_view(G, "cls.kernel")[0, 0]),
                                              (self.head_dw[:, 2:], p.grad_view(G, "reg.kernel")[0, 0]),
                                              (self.head_db[:2], p.grad_view(G, "cls.bias")),
                                              (self.head_db[2:], p.grad_view(G, "reg.bias"))], self.device)
            if not self.compose_head:
                self._up_bias_split = ops.CopyTable(
                    [(self.up_db[256 * L["slot"]:256 * (L["slot"] + 1)], p.grad_view(G, L["conv"].bias))
                     for L in self.layers if L["kind"] == "deconv"], self.device)

        def head_leaves():
            ops.conv_wgrad(self.head_geom, a["concat"], d["head"], self.head_dw, self.wgrad_ws)
            ops.colsum(d["head"], 16, M, 16, self.head_db, ws_tag="side")
            self._head_split.run()

        def concat_leaves():
            # the three deconv bias gradients are the column sums of the concat gradient: one pass over it
            ops.colsum(d["concat"], 768, M, 768, self.up_db, ws_tag="side")
            self._up_bias_split.run()

        if self.compose_head:
            # collapsed branches + heads: the head gradient feeds the three 16-channel contractions directly; its column
            # sums (the heads' bias gradient, and through H the branch biases) are the only pass over it
            self._dshuffle.run(d["head"], backward=True)
            # (queued, not flushed: the leaves and branches that hang off the head gradient cross to the second stream
            # behind ONE event, with the first of them that is issued through on_side below)
            if not skip_leaves:
                pending.append((lambda: ops.colsum(d["head"], 16, M, 16, self.head_db, ws_tag="side"), False))
        else:
            on_side(head_leaves)
            ops.conv_forward(self.head_dgeom, d["head"], self.packed_t["head"][0], d["concat"])
            on_side(concat_leaves)
        layers = self.layers
        branches_left = [len(DECONVS)]
        batched_convs = set().union(*[names for _, names in self.wgrad_batches.values()]) if self.wgrad_batches else set()
        first_write = set()                    # gradient buffers that already hold a contribution

        # ---- RPN blocks, last to first -------------------------------------------------------------
        writes = {}                            # gradient buffer -> contributions stored so far
        bwd_ready = {}                         # gradient buffer -> partial rows of its BN-backward statistics

        early_dst = {}                         # gradient buffer -> event behind a contribution made on the second stream
        fused_dense = {}                       # middle block -> backward sink of a Dense data gradient that rode on a tile
        late_reduces = []                      # slab sums of the carried Dense weight gradients (dense_dw_late)

        def dgrad_into(c, dy, dst_name, ws_tag="main", fold=None):
            ev = early_dst.pop(dst_name, None) if ws_tag == "main" else None
            if ev is not None:
                self._wait(ev, main)           # the branch's contribution is stored before this one accumulates onto it
            flags = ops.ACCUMULATE if dst_name in first_write else 0
            # the output of a middle block went through Dense(relu) (model_training.py:195): its gradient is gated
            # by that activation while the data gradient is stored (single consumer, so no ACCUMULATE there)
            mask = a[dst_name] if dst_name.endswith(".u") else None
            # the LAST contribution to the gradient of a conv output also reduces the statistics its
            # BatchNormalization backward needs (pass 1 of bn_backward folded into the store)
            writes[dst_name] = writes.get(dst_name, 0) + 1
            bwd = sink = tail = None
            if dst_name in self.bn_of and writes[dst_name] == self.consumers[dst_name]:
                bn_name, C = self.bn_of[dst_name]
                bwd, sink = (a[dst_name], self.bnstate[bn_name], True), self._bwd_sink(bn_name, C, a[dst_name].numel() // C)
                bwd_ready[dst_name] = sink
            use_w = c.name in self.packed_wu_t and fold is None
            if mask is not None and self.fuse_dense_bwd and self._tail_supported(c, dst_name, use_w):
                # the Dense(64, relu) of the block BELOW (model_training.py:195) rides on this tile: its data gradient
                # dz = (gated gradient) @ Wd^T and the statistics of the BatchNormalization under it come out of the same
                # launch (lisec_conv_extras.tail_w); the separate Dense data-gradient launch is skipped further down
                n = dst_name[:-2]
                Ln = next(L for L in self.layers if L["name"] == n)
                cn, dn = Ln["conv"], Ln["dense"]
                sink = self._bwd_sink(cn.bn, 64, cn.M)
                bwd = (a[n + ".y"], self.bnstate[cn.bn], False)
                tail = (self.packed_t[dn.name][0], d[n + ".z"])
                fused_dense[n] = sink
            if use_w:
                ops.conv_forward_winograd(self.dgeom[c.name], dy, self.packed_wu_t[c.name], d[dst_name], flags=flags,
                                          out_mask=mask, bwd=bwd, sink=sink, tail=tail)
            else:
                ops.conv_forward(self.dgeom[c.name], dy, self.packed_t[c.name][0], d[dst_name], flags=flags, out_mask=mask,
                                 bwd=bwd, sink=sink, ws_tag=ws_tag, tail=tail, fold=fold)
            first_write.add(dst_name)

        def branch_dy(L):
            """The gradient a branch's contraction produced: a concat slice, or (collapsed form) the head gradient / its
            (tap, j) columns."""
            if not self.compose_head:
                return d["concat"][:, :, 256 * L["slot"]:]
            return d["head"] if L["dT"] is None else L["dT"]

        def deconv_wgrad(L):
            c = L["conv"]
            if self.compose_head:
                b, (ts, cs) = L["slot"], L["wc_strides"]
                ops.conv_wgrad(c.g, a[L["src"]], branch_dy(L), L["G"], self.wgrad_ws, in_bn=self.bnstate[c.in_bn],
                               flags=ops.IN_RELU)
                ops.head_compose_backward(L["G"], ts, cs, p.view(L["up_kernel"]), p.view(L["up_bias"]),
                                          self.head_w[256 * b:256 * (b + 1)],
                                          self.head_db, L["k"] * L["k"], L["cin"], 256, p.grad_view(G, L["up_kernel"]),
                                          p.grad_view(G, L["up_bias"]), self.head_dw[256 * b:256 * (b + 1)])
                branches_left[0] -= 1
                if branches_left[0] == 0:
                    self._head_split.run()         # every row of dH is final: merged (768,16) -> the Keras-shaped slots
                return
            dy = d["concat"][:, :, 256 * L["slot"]:]
            if "wgeom" in L:
                ops.conv_wgrad(L["wgeom"], dy, a[L["src"]], p.grad_view(G, c.wname), self.wgrad_ws,
                               flags=ops.DY_RELU, dy_bn=self.bnstate[c.in_bn])
            else:
                ops.conv_wgrad(c.g, a[L["src"]], dy, p.grad_view(G, c.wname), self.wgrad_ws,
                               in_bn=self.bnstate[c.in_bn], flags=ops.IN_RELU, transpose_out=True)

        # the Conv2DTranspose branches of blocks 1 and 2 hang off the concat gradient, which is complete now: both of
        # their gradients go to the second stream at once, beside the small layers of blocks 3 and 2, instead of waiting
        # on the chain for their turn; the chain picks their contribution up where it reaches the block's last conv
        early_layers = set()
        if self.branch_overlap:
            for L in layers:
                if L["kind"] == "deconv" and L["slot"] < len(DECONVS) - 1:
                    ev = self._event("bwd_branch%d" % L["slot"])

                    def branch(L=L, ev=ev):
                        deconv_wgrad(L)
                        dgrad_into(L["conv"], branch_dy(L), L["src"], ws_tag="side")
                        self._record(ev, self.side)
                    pending.append((branch, False))
                    if not self.compose_head:
                        flush_side()
                    early_dst[L["src"]] = ev
                    early_layers.add(L["name"])

        for L in reversed(layers):
            c = L["conv"]
            self._mark("bwd:before " + L["name"])
            if L["kind"] == "deconv":
                if L["name"] in early_layers:
                    continue
                if self.chain_first:
                    # the chain's contraction goes into its queue BEFORE the leaves that hang off the same gradient: the
                    # second stream's queue is served first (priority) and its kernels fill every CU's LDS, so a chain
                    # kernel enqueued behind them waited for the whole leaf sequence (r03 timeline: 214 us)
                    mark_fork()
                    dgrad_into(c, branch_dy(L), L["src"])
                    on_side(lambda L=L: deconv_wgrad(L))
                else:
                    on_side(lambda L=L: deconv_wgrad(L))
                if side_filler is not None:
                    pending.append((side_filler, False))
                    flush_side()
                    side_filler = None
                if not self.chain_first:
                    dgrad_into(c, branch_dy(L), L["src"])
            elif L["kind"] == "conv":
                dst = L["dst"]
                C = c.g.Cout
                is_first_rpn = L["name"] == "rpn1.conv0"
                fold = None
                grad_w = d[dst]                        # what the weight gradient of this layer contracts against
                if dst in bwd_ready and self.fold_bn_bwd and dst in self.dyb and self._fold_supported(c, dst):
                    # dgamma / dbeta / coefficients were finalised inside the data-gradient call that stored d[dst]; the apply
                    # pass rides on the load of this layer's data gradient (no launch on the chain), and runs as a launch of
                    # the second stream into a buffer of its own for the weight gradient
                    coef = bwd_ready.pop(dst).coef
                    fold = (a[dst], self.bnstate[c.bn], coef, True)
                    grad_w = self.dyb[dst]
                    if not skip_leaves:
                        pending.append((lambda dst=dst, C=C, c=c, coef=coef, grad_w=grad_w: ops.bn_backward_apply_coef(
                            d[dst], C, a[dst], self.bnstate[c.bn], c.M, C, True, coef, grad_w), False))
                elif dst in bwd_ready:
                    ops.bn_backward_apply_coef(d[dst], C, a[dst], self.bnstate[c.bn], c.M, C, True,
                                               bwd_ready.pop(dst).coef, d[dst])
                else:
                    ops.bn_backward(d[dst], C, a[dst], self.bnstate[c.bn], c.M, C, True,
                                    p.grad_view(G, c.bn + ".gamma"), p.grad_view(G, c.bn + ".beta"), d[dst])
                if fold is None and dst in self.dyb and L["name"] in batched_convs:
                    # (the batched weight gradients were built over the dyb buffers: hand them the gradient applied in place)
                    pending.append((lambda dst=dst: self.dyb[dst].copy_(d[dst]), True))
                # the bias of a conv feeding a training-mode BN has gradient sum(dy) == 0 identically (BN removes
                # the mean); Keras' autograd returns rounding noise there -- the exact 0 stays in self.grad
                if L["name"] in batched_convs:
                    # one launch for the block's stride-1 convolutions, issued when the LAST of their output gradients
                    # (conv1's: the layers are walked back to front) is final
                    if L["name"] in self.wgrad_batches:
                        on_side(lambda batch=self.wgrad_batches[L["name"]][0]: batch.run(self.wgrad_ws))
                else:
                    on_side(lambda L=L, c=c, grad_w=grad_w: ops.conv_wgrad(
                        c.g, a[L["src"]], grad_w, p.grad_view(G, c.wname), self.wgrad_ws,
                        in_bn=self.bnstate[c.in_bn] if c.in_bn else None, flags=ops.IN_RELU if c.in_relu else 0))
                if is_first_rpn and rpn_grads_ready is not None:
                    lo = p.offsets["rpn1.conv0.kernel"][1]
                    on_side(lambda lo=lo: rpn_grads_ready(lo, p.n_theta), torch_ops=True)
                dgrad_into(c, d[dst], L["src"], fold=fold)
                if L["src"] == "fold":
                    # back through Permute + Reshape, gated by the ReLU of the last middle block's Dense (:195)
                    ops.fold_depth(d["fold"], d[self.fold_src], self.dprime, self.H * self.W, 64, inverse=True,
                                   mask=a[self.fold_src])
            else:   # mid layer: conv3d -> BN -> Dense(relu)
                n, dn = L["name"], L["dense"]
                dense_wg = lambda n=n, dn=dn: ops.conv_wgrad(dn.g, a[n + ".y"], d[n + ".u"], p.grad_view(G, dn.wname),
                                                             self.wgrad_ws, in_bn=self.bnstate[dn.in_bn])
                # the Dense weight gradient (HBM-bound, 52 granules of LDS) finds no room beside three data-gradient
                # workgroups per CU and waited 474 us in the queue IN FRONT of the block's ring weight gradient: behind it
                # (dense_wgrad_late) the ring kernel starts as soon as its gradient exists
                late_dense = self.dense_wgrad_late and self.mid_wgrad_first and L["src"] != "grid"
                # the Dense data gradient below reads both operands of the Dense weight gradient: it carries it (dense_dw)
                carried = n not in fused_dense and n in self.dense_dw_slabs
                if carried:
                    late_dense = False
                elif not late_dense:
                    on_side(dense_wg)
                # Dense data gradient; its store also reduces the statistics of the BatchNormalization under it
                if n in fused_dense:
                    msink = fused_dense.pop(n)         # done inside the data gradient of the block above (dgrad_into)
                else:
                    msink = self._bwd_sink(c.bn, 64, c.M)
                    ops.conv_forward(self.dgeom[dn.name], d[n + ".u"], self.packed_t[dn.name][0], d[n + ".z"],
                                     bwd=(a[n + ".y"], self.bnstate[c.bn], False), sink=msink,
                                     dense_dw=self.dense_dw_slabs[n] if carried else None)
                    if carried:
                        reduce = lambda n=n, dn=dn: ops.dense_dw_reduce(self.dense_dw_slabs[n], p.grad_view(G, dn.wname))
                        # the 8 MB slab sum finds no registers beside the Winograd workgroups (46 - 60 us in the step for 6 us of
                        # work) and the second stream is in order: enqueued right here it held the block's weight gradient back;
                        # nothing reads the result before the optimizer, so the three sums go behind the last weight gradient
                        if self.dense_dw_late:
                            late_reduces.append(reduce)
                        else:
                            on_side(reduce)
                if L["src"] != "grid":
                    ops.bn_backward_apply_coef(d[n + ".z"], 64, a[n + ".y"], self.bnstate[c.bn], c.M, 64, False, msink.coef,
                                               d[n + ".z"])
                if L["src"] == "grid":
                    # the grid is a constant on the empty cells + V voxel rows: both gradients reduce to V-row
                    # contractions plus sums of dy over boundary-trimmed boxes (exact; csrc/sparse_grid.hip)
                    rows = (sample.coords, sample.info, rcap)
                    dg = self.dgeom[c.name]
                    dW = p.grad_view(G, c.wname)
                    # the apply pass of this block's BatchNormalization backward runs inside the line sums (one pass
                    # over the 82 MB gradient instead of two)
                    ops.tap_sums_bn(c.g, d[n + ".z"], a[n + ".y"], self.bnstate[c.bn], msink.coef, d[n + ".z"], self.mid1_S,
                                    self.tapsum_ws)
                    ops.const_field_grads(p.view(c.wname), self.mid1_S, None, 27, 64, 64, g_all=self.g_all)
                    vout, delta = self.vfe.saved_field("vout"), self.vfe.saved_field("delta")

                    def sparse_wgrad(dg=dg, dW=dW, dz=d[n + ".z"], rows=rows, vout=vout, delta=delta):
                        ops.conv_wgrad(dg, dz, delta, dW, self.wgrad_ws, transpose_out=True, rows=rows)
                        ops.const_field_grads(None, self.mid1_S, vout, 27, 64, 64, dW=dW, cvec_row=sample.info,
                                              cvec_row_max=sample.cap)
                    on_side(sparse_wgrad)
                    ops.conv_forward(dg, d[n + ".z"], self.packed_t[c.name][0], self.dout_rows, rows=rows,
                                     queue=self.rows_queue)
                else:
                    # the data gradient FIRST, the weight gradient behind it on the second stream: both fill the chip
                    # on their own and run slower side by side than one after the other (mid2: 800 us together,
                    # 333 + 358 alone); behind the data gradient the weight gradient shares the chip with the
                    # short kernels of the rest of the chain instead
                    if c.name in self.wino_wgrad_ws:
                        # Winograd-domain weight gradient (csrc/wino_wgrad.hip): 4 / 9 of the ring kernel's MFMAs
                        wg = lambda L=L, c=c, n=n: ops.conv_wgrad_winograd(c.g, a[L["src"]], d[n + ".z"],
                                                                           p.grad_view(G, c.wname), self.wino_wgrad_ws[c.name])
                    else:
                        wg = lambda L=L, c=c, n=n: ops.conv_wgrad(c.g, a[L["src"]], d[n + ".z"], p.grad_view(G, c.wname),
                                                                  self.wgrad_ws)
                    if self.mid_wgrad_first:
                        on_side(wg)
                        if late_dense:
                            pending.append((dense_wg, False))
                            flush_side()
                        dgrad_into(c, d[n + ".z"], L["src"])
                    else:
                        dgrad_into(c, d[n + ".z"], L["src"])
                        on_side(wg)
        # ---- VFE -----------------------------------------------------------------------------------
        self._mark("bwd:before vfe")
        for fn in late_reduces:
            pending.append((fn, False))
        flush_side()
        self.vfe.backward(None, G, dout_rows=self.dout_rows, g_all=self.g_all)
        if self._join_event is None:
            self._join_event = self._new_event()
        self._mark("bwd:vfe done")
        self._record(self._join_event, self.side)
        self._wait(self._join_event, main)     # every weight gradient has landed before the optimizer reads G
        self._mark("bwd:joined")
        return self.loss_out

    def early_update(self, lo, hi, lr=0.01, decay=1e-6, momentum=0.9):
        """SGD-Nesterov of theta[lo:hi] AHEAD of the rest of the step (backward's rpn_grads_ready hook, on the second stream):
        the RPN + head variables -- 94 % of the parameters -- have final gradients while the middle layers and the VFE are
        still being differentiated, and nothing in the rest of the backward pass reads theta itself (the contractions read
        the packed copies), so their 26 MB update runs under the MFMA-bound kernels instead of at the serial end of the
        step.  Elementwise, hence the same values whichever call updates an element.  apply_gradients() then updates
        theta[:lo] and advances the iteration count."""
        if not self.early_sgd or lo % 4 or hi != self.params.n_theta:
            return
        p = self.params
        n = (hi - lo) // 4 * 4
        ops.sgd_nesterov_step_dev(p.theta[lo:lo + n], self.grad[lo:lo + n], self.velocity[lo:lo + n], lr, decay, momentum,
                                  self._iter_dev, advance=False)
        self._early_from = lo

    def apply_gradients(self, lr=0.01, decay=1e-6, momentum=0.9):
        """optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True) (model_training.py:295)."""
        # lr_t = lr / (1 + decay * iterations), derived on the device from its own iteration counter
        lo = getattr(self, "_early_from", None)
        self._early_from = None
        if lo is not None:
            # the tail of the buffer was updated by early_update() during the backward pass
            ops.sgd_nesterov_step_dev(self.params.theta[:lo], self.grad[:lo], self.velocity[:lo], lr, decay, momentum,
                                      self._iter_dev)
        else:
            ops.sgd_nesterov_step_dev(self.params.theta, self.grad, self.velocity, lr, decay, momentum, self._iter_dev)
        self._mark("step:updated")
        self._iterations += 1
        self.params_version += 1
        if self._train_ready and self.early_pack:
            # both repacks (forward and transposed layouts, ~75 us) for the NEXT step go to the second stream now: they
            # only depend on this update, and the next sweep's voxeliser + VFE (~105 us) do not read them
            if getattr(self, "_pack_done", None) is None:
                self._pack_fork, self._pack_done, self._pack_late = self._new_event(), self._new_event(), self._new_event()
            self._record(self._pack_fork, torch.cuda.current_stream())
            self._wait(self._pack_fork, self.side)
            pin = _lib.pin_stream(self.side.cuda_stream)
            try:
                self._pack_all(after_main=lambda: self._record(self._pack_done, self.side))
                self._pack_all_t()
            finally:
                _lib.pin_stream(pin)
            self._record(self._pack_late, self.side)
            self._pack_pending = True
            self._late_pending = True

    def train_step(self, sample, y_cls, y_reg, loss="mse", allreduce=None):
        """One fit() step at batch_size=1: forward (batch statistics) + backward + SGD-Nesterov.
        allreduce: optional callable(grad) that averages the flat gradient across data-parallel ranks."""
        self.forward(sample, training=True)
        if allreduce is not None and hasattr(allreduce, "start_tail"):
            # two buckets: the RPN + head gradients (the tail of theta) are reduced under the rest of the backward
            self.backward(y_cls, y_reg, loss=loss, rpn_grads_ready=lambda lo, hi: allreduce.start_tail(self.grad, lo, hi))
            allreduce.finish(self.grad)
        elif allreduce is not None:
            self.backward(y_cls, y_reg, loss=loss)
            allreduce(self.grad)
        else:
            self.backward(y_cls, y_reg, loss=loss, rpn_grads_ready=lambda lo, hi: self.early_update(lo, hi))
        self.apply_gradients()
        return self.loss_out


class StalePlanError(RuntimeError):
    """A recorded step plan holds raw addresses of buffers that an eager call has since reallocated."""


def _check_plan_fresh(step):
    if step.alloc_gen != _lib.alloc_generation():
        raise StalePlanError(
            "this step plan was recorded before a workspace of the network / VFE / voxeliser was reallocated (an eager "
            "call on a larger sweep or grid): replaying it would write through freed addresses.  Record a new one "
            "(Model.fit does so by itself).")


def _sync_packed(net):
    """The recorded forward contains no repack (the previous step's update left one pending).  If the variables were
    changed since (params.load_dict / touch, an eager step), repack them now on the replay stream and re-arm the two events
    the recorded forward waits for."""
    cur = (net.params_version, net.params.version)
    if net._packed_version != cur or net._packed_t_version != cur or not getattr(net, "_pack_pending", False):
        net._pack_pending = False
        net._late_pending = False
        net._pack_all()
        net._pack_all_t()
        net._record(net._pack_done, torch.cuda.current_stream())
        net._record(net._pack_late, torch.cuda.current_stream())
        net._pack_pending = True
        net._late_pending = True


class RecordedStep:
    """One whole fit() step -- voxelise, forward, backward (both streams, fork / join events included), SGD-Nesterov, the
    weight repack for the next step -- recorded ONCE as a step plan of the C ABI (lisec_step_plan_*, csrc/plan.hip) and
    re-issued by one C call per step: the ~250 launches cost the host one ctypes call instead of ~1.5 ms of Python
    (schedule, plan selection, argument marshalling).  It IS the eager schedule -- same kernels, same streams, same
    events, bit-identical variables -- not a HIP graph (a captured graph of this two-stream step replays 2x slower than
    the eager launches on ROCm 7.2).  The schedule is static; what varies from sample to sample lives in device memory:

      points   a fixed-capacity (capacity, 3) buffer; a sweep with fewer points is padded with points far outside
               the grid, which the voxeliser's range test (model_training.py:118-120) drops -- kept points, their
               order and therefore every voxel and feature row are exactly those of the unpadded sweep
      targets  (Ho,Wo,2) / (Ho,Wo,14) static buffers
      lr_t     derived by the SGD kernel from the device iteration counter (lisec_sgd_nesterov_step_dev)

    Record and replay on ONE torch stream (the current stream at construction).  Data parallel (allreduce=): the gradient
    exchange is part of the plan."""

    PAD = 1.0e6          # metres: floor(1e6 / 0.5) is far beyond maxVoxelX, the point is dropped like any other outlier

    def __init__(self, net, voxelizer, capacity, dtype=torch.float32, loss="mse", lr=0.01, decay=1e-6, momentum=0.9,
                 warmup=2, allreduce=None):
        import ctypes
        self.net, self.vox, self.capacity, self.loss = net, voxelizer, int(capacity), loss
        # data parallel: the two-bucket gradient exchange (parallel._BucketedAverage) is part of the recorded schedule --
        # lisec_allreduce_grads and its event edges record themselves, a torch.distributed exchange rides as host calls.
        # Every rank records and replays the same sequence (the warm-up and recording steps exchange gradients for real).
        self.allreduce = allreduce
        dev = net.device
        self.lib = _lib.load()
        self.points = torch.full((self.capacity, 3), self.PAD, dtype=dtype, device=dev)
        self.ycls = torch.zeros((net.Ho, net.Wo, 2), dtype=torch.float32, device=dev)
        self.yreg = torch.zeros((net.Ho, net.Wo, 14), dtype=torch.float32, device=dev)
        self.hyper = (lr, decay, momentum)
        self.sample = None
        self.stream_handle = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize(dev)
        net._prepare_training()
        p = net.params
        keep = (p.theta.clone(), p.state.clone(), net.velocity.clone(), net._iter_dev.clone(), net._iterations)
        # eager warm-up (lazy workspaces, descriptor tables, events; it leaves the next step's repack pending, which is
        # the state every recorded step starts from), then one more step that is recorded while it runs
        for _ in range(max(1, warmup)):
            self._enqueue()
        torch.cuda.synchronize(dev)
        self.plan = ctypes.c_void_p()
        _lib.check(self.lib.lisec_step_plan_create(ctypes.byref(self.plan)))
        _lib.check(self.lib.lisec_step_plan_begin(self.plan))
        try:
            self._enqueue()
        finally:
            _lib.check(self.lib.lisec_step_plan_end(self.plan))
        torch.cuda.synchronize(dev)
        self.launches = self.lib.lisec_step_plan_size(self.plan)
        # those steps trained on the padding: put every variable back and repack the kernels from them
        p.theta.copy_(keep[0]); p.state.copy_(keep[1]); net.velocity.copy_(keep[2]); net._iter_dev.copy_(keep[3])
        net._iterations = keep[4]
        net.params_version += 1
        net.state_version += 1
        p.touch()
        net._pack_pending = False
        net._pack_all()
        net._pack_all_t()
        net._record(net._pack_done, torch.cuda.current_stream())   # what the recorded forward waits for
        net._record(net._pack_late, torch.cuda.current_stream())
        net._pack_pending = True
        net._late_pending = True
        torch.cuda.synchronize(dev)
        self.alloc_gen = _lib.alloc_generation()

    def _enqueue(self):
        net = self.net
        self.sample = self.vox(self.points, out=self.sample)
        net.forward(self.sample, training=True)
        ar = self.allreduce
        if ar is not None and hasattr(ar, "start_tail"):
            net.backward(self.ycls, self.yreg, loss=self.loss,
                         rpn_grads_ready=lambda lo, hi: ar.start_tail(net.grad, lo, hi))
            ar.finish(net.grad)
        elif ar is not None:
            net.backward(self.ycls, self.yreg, loss=self.loss)
            ar(net.grad)
        else:
            net.backward(self.ycls, self.yreg, loss=self.loss,
                         rpn_grads_ready=lambda lo, hi: net.early_update(lo, hi, *self.hyper))
        net.apply_gradients(*self.hyper)

    def _check_stream(self):
        if torch.cuda.current_stream().cuda_stream != self.stream_handle:
            raise RuntimeError("a RecordedStep replays on the stream it was recorded on: make that stream current")

    def load(self, points, ycls, yreg):
        """Stage one sweep: points (n <= capacity, >= 3 columns; device or host tensor / numpy), targets (Ho,Wo,2|14)."""
        self._check_stream()
        pts = torch.as_tensor(points)
        n = int(pts.shape[0])
        if n > self.capacity:
            raise ValueError(f"sweep of {n} points exceeds the recorded capacity {self.capacity}")
        self.points[:n].copy_(pts[:, :3], non_blocking=True)
        if n < self.capacity:
            self.points[n:].fill_(self.PAD)
        self.ycls.copy_(torch.as_tensor(ycls).reshape(self.ycls.shape), non_blocking=True)
        self.yreg.copy_(torch.as_tensor(yreg).reshape(self.yreg.shape), non_blocking=True)

    def replay(self):
        """Runs the recorded step on what load() staged; returns net.loss_out (device, [total, class, regression])."""
        self._check_stream()
        _check_plan_fresh(self)
        net = self.net
        _sync_packed(net)
        _lib.check(self.lib.lisec_step_plan_run(self.plan))
        net._iterations += 1
        net.params_version += 1          # theta moved; the recorded step also repacked it for the next one
        net.state_version += 1
        net._packed_version = net._packed_t_version = (net.params_version, net.params.version)
        net._pack_pending = True
        net._late_pending = True
        self.sample._host_info = None
        return net.loss_out

    def __call__(self, points, ycls, yreg):
        self.load(points, ycls, yreg)
        return self.replay()

    def close(self):
        if getattr(self, "plan", None):
            torch.cuda.synchronize(self.net.device)
            self.lib.lisec_step_plan_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelinedStep:
    """RecordedStep with the input pipeline folded in: step k voxelises the sweep of step k + 1 on the second stream, in the
    ~200 us that stream idles at the start of the backward pass, instead of step k + 1 starting with 90 us of seven small
    dependent launches in front of its first contraction.  Two sets of (points, targets, voxel sample) buffers alternate,
    so there are two recorded plans; the variables after every step are bit-identical to the eager schedule's (a sweep's
    voxels do not depend on when they are computed).

        step = PipelinedStep(net, voxelizer, capacity)
        step.prime(points0, ycls0, yreg0)                    # stage + voxelise the first sweep
        for k in range(n):
            loss = step.step(points[k + 1], ycls[k + 1], yreg[k + 1])     # trains on sweep k, prepares sweep k + 1
                                                                          # (no arguments: the staged buffers are reused)
    """
    PAD = RecordedStep.PAD

    def __init__(self, net, voxelizer, capacity, dtype=torch.float32, loss="mse", lr=0.01, decay=1e-6, momentum=0.9,
                 warmup=2, allreduce=None):
        import ctypes
        self.net, self.vox, self.capacity, self.loss = net, voxelizer, int(capacity), loss
        self.allreduce = allreduce                 # data parallel: see RecordedStep
        dev = net.device
        self.lib = _lib.load()
        self.points = [torch.full((self.capacity, 3), self.PAD, dtype=dtype, device=dev) for _ in range(2)]
        self.ycls = [torch.zeros((net.Ho, net.Wo, 2), dtype=torch.float32, device=dev) for _ in range(2)]
        self.yreg = [torch.zeros((net.Ho, net.Wo, 14), dtype=torch.float32, device=dev) for _ in range(2)]
        self.hyper = (lr, decay, momentum)
        self.stream_handle = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize(dev)
        net._prepare_training()
        p = net.params
        keep = (p.theta.clone(), p.state.clone(), net.velocity.clone(), net._iter_dev.clone(), net._iterations)
        self.samples = [self.vox(self.points[j]) for j in range(2)]
        for k in range(2 * max(1, warmup)):
            self._enqueue(k & 1)
        torch.cuda.synchronize(dev)
        self.plans = []
        for j in range(2):
            plan = ctypes.c_void_p()
            _lib.check(self.lib.lisec_step_plan_create(ctypes.byref(plan)))
            _lib.check(self.lib.lisec_step_plan_begin(plan))
            try:
                self._enqueue(j)
            finally:
                _lib.check(self.lib.lisec_step_plan_end(plan))
            self.plans.append(plan)
        torch.cuda.synchronize(dev)
        self.launches = self.lib.lisec_step_plan_size(self.plans[0])
        p.theta.copy_(keep[0]); p.state.copy_(keep[1]); net.velocity.copy_(keep[2]); net._iter_dev.copy_(keep[3])
        net._iterations = keep[4]
        net.params_version += 1
        net.state_version += 1
        p.touch()
        net._pack_pending = False
        net._pack_all()
        net._pack_all_t()
        net._record(net._pack_done, torch.cuda.current_stream())   # what the recorded forward waits for
        net._record(net._pack_late, torch.cuda.current_stream())
        net._pack_pending = True
        net._late_pending = True
        self.cur = 0
        torch.cuda.synchronize(dev)
        self.alloc_gen = _lib.alloc_generation()

    def _enqueue(self, j):
        net = self.net
        net.forward(self.samples[j], training=True)
        filler = lambda: self.vox(self.points[1 - j], out=self.samples[1 - j])
        ar = self.allreduce
        if ar is not None and hasattr(ar, "start_tail"):
            net.backward(self.ycls[j], self.yreg[j], loss=self.loss, side_filler=filler,
                         rpn_grads_ready=lambda lo, hi: ar.start_tail(net.grad, lo, hi))
            ar.finish(net.grad)
        elif ar is not None:
            net.backward(self.ycls[j], self.yreg[j], loss=self.loss, side_filler=filler)
            ar(net.grad)
        else:
            net.backward(self.ycls[j], self.yreg[j], loss=self.loss, side_filler=filler,
                         rpn_grads_ready=lambda lo, hi: net.early_update(lo, hi, *self.hyper))
        net.apply_gradients(*self.hyper)

    def _check_stream(self):
        if torch.cuda.current_stream().cuda_stream != self.stream_handle:
            raise RuntimeError("a PipelinedStep replays on the stream it was recorded on: make that stream current")

    def _load(self, j, points, ycls, yreg):
        pts = torch.as_tensor(points)
        n = int(pts.shape[0])
        if n > self.capacity:
            raise ValueError(f"sweep of {n} points exceeds the recorded capacity {self.capacity}")
        self.points[j][:n].copy_(pts[:, :3], non_blocking=True)
        if n < self.capacity:
            self.points[j][n:].fill_(self.PAD)
        self.ycls[j].copy_(torch.as_tensor(ycls).reshape(self.ycls[j].shape), non_blocking=True)
        self.yreg[j].copy_(torch.as_tensor(yreg).reshape(self.yreg[j].shape), non_blocking=True)

    def prime(self, points, ycls, yreg):
        """Stages the FIRST sweep and voxelises it (outside the plans); the next step() trains on it."""
        self._check_stream()
        self._load(self.cur, points, ycls, yreg)
        self.vox(self.points[self.cur], out=self.samples[self.cur])

    def stage_next(self, points, ycls, yreg):
        """Stages the sweep the next step() voxelises (and the step() after it trains on)."""
        self._check_stream()
        self._load(1 - self.cur, points, ycls, yreg)

    def step(self, next_points=None, next_ycls=None, next_yreg=None):
        """Trains on the current sweep and voxelises the staged next one; returns net.loss_out (device)."""
        self._check_stream()
        if next_points is not None:
            self._load(1 - self.cur, next_points, next_ycls, next_yreg)
        _check_plan_fresh(self)
        net = self.net
        _sync_packed(net)
        _lib.check(self.lib.lisec_step_plan_run(self.plans[self.cur]))
        net._iterations += 1
        net.params_version += 1
        net.state_version += 1
        net._packed_version = net._packed_t_version = (net.params_version, net.params.version)
        net._pack_pending = True
        net._late_pending = True
        for s_ in self.samples:
            s_._host_info = None
        self.cur ^= 1
        return net.loss_out

    def close(self):
        if getattr(self, "plans", None):
            torch.cuda.synchronize(self.net.device)
            for plan in self.plans:
                self.lib.lisec_step_plan_destroy(plan)
            self.plans = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
