"""Host side of the sparse-exact VFE stack (lisec_vfe_forward/backward, include/lisec_hip.h section 2).

Stands where the reference applies addVFELayer(6,32), addVFELayer(32,64), addFCN(64,64) and
MaxPoolingVFELayer(combine=True) to the dense (D,H,W,T,6) tensor (model_training.py:231-235).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import VfeGrads, VfeParams

VFE_LAYERS = ("vfe1", "vfe2", "fcn")


class VFEStack:
    def __init__(self, params, device=None):
        self.device = device or _lib.require_gpu()
        self.lib = _lib.load()
        self.params = params
        self._ws = torch.empty(self.lib.lisec_vfe_workspace_bytes(), dtype=torch.uint8, device=self.device)
        self._saved = None
        self._sample = None
        # LISEC_VFE_BWD=valu: the row-by-row backward of layers 3 and 2 instead of the 32-row MFMA tiles
        self.tiled = _lib.knob("vfe_bwd_tiled", True)
        self.tiled_min_points = _lib.knob("vfe_tiled_min_points", 65536)
        self._saved_rows = 0

    def _cparams(self, theta=None):
        p, s = self.params, VfeParams()
        for i, n in enumerate(VFE_LAYERS):
            s.kernel[i] = p.ptr(f"{n}.dense.kernel", theta).value
            s.gamma[i] = p.ptr(f"{n}.bn.gamma", theta).value
            s.beta[i] = p.ptr(f"{n}.bn.beta", theta).value
            s.moving_mean[i] = p.ptr(f"{n}.bn.moving_mean").value
            s.moving_var[i] = p.ptr(f"{n}.bn.moving_variance").value
        return s

    def forward(self, sample, training, out=None, dense=True):
        """sample: VoxelSample.  Returns the dense (D, H, W, 64) grid (device tensor); dense=False: no grid, only the
        compact per-voxel outputs (saved_field('vout') / ('delta')) that the field form of the first Conv3D reads."""
        D, H, W = sample.grid_shape
        ncells = D * H * W
        # training with the tiled backward: `saved` also carries the per-row extras (winner slots, layer-2 rows).  The
        # tiles pay off on big sweeps (a 200 k-point sweep: 0.6 -> 0.3 ms); on a 20 k-point one the row-by-row kernels
        # are as fast and need two launches fewer, so the choice follows the sweep's size (known on the host)
        rows_cap = sample.n_points if (training and self.tiled and sample.n_points >= self.tiled_min_points) else 0
        need = (self.lib.lisec_vfe_saved_floats_rows(sample.cap, rows_cap) if rows_cap
                else self.lib.lisec_vfe_saved_floats(sample.cap))
        if self._saved is None or self._saved.numel() < need:
            self._saved = torch.empty(need, dtype=torch.float32, device=self.device)
            _lib.bump_alloc_generation()           # recorded step plans hold the old address
        self._saved_rows = rows_cap
        grid = None
        if dense:
            grid = out if out is not None else torch.empty((D, H, W, 64), dtype=torch.float32, device=self.device)
        cp = self._cparams()
        _lib.check(self.lib.lisec_vfe_forward(
            ctypes.byref(cp), _lib.ptr(sample.info), _lib.ptr(sample.cell_voxel), _lib.ptr(sample.npts),
            _lib.ptr(sample.row_start), _lib.ptr(sample.rows), _lib.ptr(getattr(sample, "row_stats", None)),
            rows_cap, ncells, sample.cfg.sampleSize, sample.cap, 1 if training else 0, _lib.ptr(self._saved), _lib.ptr(self._ws), self._ws.numel(),
            _lib.ptr(grid), _lib.current_stream()))
        self._sample = sample
        return grid

    def rewrite_grid(self, out):
        """Rewrite the dense grid of the last forward from the saved per-voxel statistics."""
        sample = self._sample
        D, H, W = sample.grid_shape
        _lib.check(self.lib.lisec_vfe_grid_from_saved(_lib.ptr(sample.info), _lib.ptr(sample.cell_voxel), D * H * W,
                                                      sample.cap, _lib.ptr(self._saved), _lib.ptr(out),
                                                      _lib.current_stream()))
        return out

    def saved_field(self, which):
        """View of the compact per-voxel outputs of the last forward: 'vout' / 'delta', shape (cap+1, 64)."""
        cap = self._sample.cap
        off = self.lib.lisec_vfe_saved_field_offset(cap, {"vout": 0, "delta": 1}[which])
        return self._saved[off:off + (cap + 1) * 64].view(cap + 1, 64)

    def backward(self, dgrid, grad, dout_rows=None, g_all=None):
        """dgrid: (D,H,W,64) gradient wrt the grid of the last training forward -- or None with the compact
        form (dout_rows (cap+1,64): gradient at the occupied cells, g_all (64,): its sum over all cells);
        grad: flat gradient buffer laid out like params.theta (VFE entries are overwritten)."""
        sample, p = self._sample, self.params
        D, H, W = sample.grid_shape
        need = self.lib.lisec_vfe_backward_workspace_bytes(sample.cap, sample.n_points)
        if getattr(self, "_bws", None) is None or self._bws.numel() < need:
            self._bws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.bump_alloc_generation()
        g = VfeGrads()
        for i, n in enumerate(VFE_LAYERS):
            g.kernel[i] = p.ptr(f"{n}.dense.kernel", grad).value
            g.gamma[i] = p.ptr(f"{n}.bn.gamma", grad).value
            g.beta[i] = p.ptr(f"{n}.bn.beta", grad).value
        cp = self._cparams()
        _lib.check(self.lib.lisec_vfe_backward(
            ctypes.byref(cp), _lib.ptr(sample.info), _lib.ptr(sample.cell_voxel), _lib.ptr(sample.npts),
            _lib.ptr(sample.row_start), _lib.ptr(sample.rows), sample.n_points, D * H * W,
            sample.cfg.sampleSize, sample.cap, _lib.ptr(self._saved), _lib.ptr(dgrid), _lib.ptr(dout_rows),
            _lib.ptr(g_all), ctypes.byref(g), 1 if self._saved_rows else 0,
            _lib.ptr(self._bws), self._bws.numel(), _lib.current_stream()))
