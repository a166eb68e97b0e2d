"""Host side of the HIP voxeliser (lisec_voxelize, include/lisec_hip.h section 1).

Mirrors VFE_preprocessing of the reference (model_training.py:112-152): same arguments,
same (z, x, y, t, f) index convention, but the result stays on the GPU as occupied voxels +
compact feature rows instead of a 134 M element dense tensor.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import VoxelCfg


class VoxelSample:
    """Device-resident result of one voxeliser call (one lidar sweep).

    info        int32[8]   V, rows, valid points, max count, overflow (LISEC_VI_*)
    cell_voxel  int32[NZ*NX*NY]   voxel ordinal per grid cell (-1 = empty)
    coords      int32[cap,3]      (z, x, y)
    counts/npts int32[cap]
    row_start   int32[cap+1]
    rows        float32[n,6]      compact feature rows
    row_point   int32[n]          original point index per row
    row_stats   int64[LISEC_ROW_STATS_WORDS]  first / second moments of the rows (fixed point) + VFE scratch
    """

    def __init__(self, cfg, n_points, cap, info, cell_voxel, coords, counts, npts, row_start, rows,
                 row_point, row_stats=None):
        self.cfg, self.n_points, self.cap = cfg, n_points, cap
        self.info, self.cell_voxel, self.coords = info, cell_voxel, coords
        self.counts, self.npts, self.row_start = counts, npts, row_start
        self.rows, self.row_point, self.row_stats = rows, row_point, row_stats
        self._host_info = None

    @property
    def grid_shape(self):
        c = self.cfg
        return (c.maxVoxelZ, 2 * c.maxVoxelX, 2 * c.maxVoxelY)

    def host_info(self):
        """Synchronises; returns dict(V, rows, valid, max_count)."""
        if self._host_info is None:
            h = self.info.cpu().numpy()
            if h[4]:
                raise _lib.LisecError("voxeliser output capacity overflow")
            self._host_info = dict(V=int(h[0]), rows=int(h[1]), valid=int(h[2]), max_count=int(h[3]))
        return self._host_info

    def to_host(self):
        """numpy copies trimmed to V / rows (same keys as oracle.voxel_ref.voxelize_ref)."""
        hi = self.host_info()
        V, R, T = hi["V"], hi["rows"], self.cfg.sampleSize
        out = dict(coords=self.coords[:V].cpu().numpy(), counts=self.counts[:V].cpu().numpy(),
                   npts=self.npts[:V].cpu().numpy(), row_start=self.row_start[:V + 1].cpu().numpy(),
                   rows=self.rows[:R].cpu().numpy(), row_point=self.row_point[:R].cpu().numpy())
        out["feats"] = self.padded_feats()[:V].cpu().numpy()
        pidx = np.full((V, T), -1, dtype=np.int32)
        for_rows = np.repeat(np.arange(V), out["npts"])
        slot = np.arange(R) - out["row_start"][for_rows]
        pidx[for_rows, slot] = out["row_point"]
        out["point_index"] = pidx
        return out

    def padded_feats(self):
        """(V, T, 6) zero padded feature blocks on the device (model_training.py:141)."""
        V = self.host_info()["V"]
        T = self.cfg.sampleSize
        padded = torch.empty((max(V, 1), T, 6), dtype=torch.float32, device=self.rows.device)
        lib = _lib.load()
        _lib.check(lib.lisec_voxel_rows_to_padded(_lib.ptr(self.info), _lib.ptr(self.npts),
                                                  _lib.ptr(self.row_start), _lib.ptr(self.rows), T, V,
                                                  _lib.ptr(padded), _lib.current_stream()))
        return padded[:V]


def host_row_stats(rows):
    """row_stats for rows that did not come from lisec_voxelize (a dense array turned into a sample): the 6 + 21
    moments in the same two-limb fixed-point layout (everything in replica 0), VFE scratch zeroed."""
    r = np.asarray(rows, dtype=np.float64).reshape(-1, 6)
    vals = [r[:, j].sum() for j in range(6)] + [(r[:, j] * r[:, k]).sum() for j in range(6) for k in range(j, 6)]
    out = np.zeros(_lib.ROW_STATS_WORDS, dtype=np.int64)
    for i, s in enumerate(vals):
        hi = np.floor(s * 256.0)
        out[2 * i] = int(hi)
        out[2 * i + 1] = int(np.rint((s - hi / 256.0) * 1099511627776.0))
    return out


class Voxelizer:
    """Reusable voxeliser for one grid; owns its workspace (grown on demand)."""

    def __init__(self, xSize, ySize, zSize, sampleSize, maxVoxelX, maxVoxelY, maxVoxelZ, device=None):
        self.device = device or _lib.require_gpu()
        self.lib = _lib.load()
        self.cfg = VoxelCfg(float(xSize), float(ySize), float(zSize), int(maxVoxelX), int(maxVoxelY),
                            int(maxVoxelZ), int(sampleSize))
        self.ncells = 4 * int(maxVoxelX) * int(maxVoxelY) * int(maxVoxelZ)
        self._ws = None

    def _workspace(self, n):
        need = self.lib.lisec_voxelize_workspace_bytes(ctypes.byref(self.cfg), n)
        if need == 0:
            raise _lib.LisecError("invalid voxel grid configuration: " + self.lib.lisec_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.bump_alloc_generation()           # recorded step plans hold the old address
        return self._ws

    def __call__(self, points, out=None):
        """points: (N, >=3) float32/float64 numpy array or torch tensor (host or device).
        out: a VoxelSample of an earlier call with the same number of points whose buffers are written again (a recorded
        step plan points at fixed addresses)."""
        if isinstance(points, np.ndarray):
            if points.dtype not in (np.float32, np.float64):
                points = points.astype(np.float64)
            points = torch.from_numpy(np.ascontiguousarray(points))
        if points.dtype not in (torch.float32, torch.float64):
            points = points.double()
        if points.dim() != 2 or points.shape[1] < 3:
            raise ValueError("points must have shape (N, >=3)")
        pts = points.to(self.device, non_blocking=True).contiguous()
        n, stride = int(pts.shape[0]), int(pts.shape[1])
        cap = min(n, self.ncells)
        dev, i32 = self.device, torch.int32
        ws = self._workspace(n)
        if out is not None:
            if out.n_points != n or out.cap != cap:
                raise ValueError("out: a sample of another size")
            _lib.check(self.lib.lisec_voxelize(
                ctypes.byref(self.cfg), _lib.ptr(pts), 0 if pts.dtype == torch.float32 else 1, n, stride,
                _lib.ptr(ws), ws.numel(), cap, _lib.ptr(out.info), _lib.ptr(out.cell_voxel), _lib.ptr(out.coords),
                _lib.ptr(out.counts), _lib.ptr(out.npts), _lib.ptr(out.row_start), _lib.ptr(out.rows),
                _lib.ptr(out.row_point), _lib.ptr(out.row_stats), _lib.current_stream()))
            out._keepalive, out._host_info = pts, None
            return out
        info = torch.empty(8, dtype=i32, device=dev)
        cell_voxel = torch.empty(self.ncells, dtype=i32, device=dev)
        coords = torch.empty((max(cap, 1), 3), dtype=i32, device=dev)
        counts = torch.empty(max(cap, 1), dtype=i32, device=dev)
        npts = torch.empty(max(cap, 1), dtype=i32, device=dev)
        row_start = torch.empty(max(cap, 1) + 1, dtype=i32, device=dev)
        rows = torch.empty((max(n, 1), 6), dtype=torch.float32, device=dev)
        row_point = torch.empty(max(n, 1), dtype=i32, device=dev)
        row_stats = torch.empty(_lib.ROW_STATS_WORDS, dtype=torch.int64, device=dev)
        _lib.check(self.lib.lisec_voxelize(
            ctypes.byref(self.cfg), _lib.ptr(pts), 0 if pts.dtype == torch.float32 else 1, n, stride,
            _lib.ptr(ws), ws.numel(), cap, _lib.ptr(info), _lib.ptr(cell_voxel), _lib.ptr(coords),
            _lib.ptr(counts), _lib.ptr(npts), _lib.ptr(row_start), _lib.ptr(rows), _lib.ptr(row_point),
            _lib.ptr(row_stats), _lib.current_stream()))
        s = VoxelSample(self.cfg, n, cap, info, cell_voxel, coords, counts, npts, row_start, rows,
                        row_point, row_stats)
        s._keepalive = pts
        return s
