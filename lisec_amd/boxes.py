"""Host side of the box-geometry kernels (include/lisec_hip.h section 5): the reference's label generation
(serialize_data.preprocessLabels) and RPN post-processing (rpnToRegion.rpnToRegion) with the same names."""
import ctypes
import math
import random

import numpy as np
import torch

from . import Constants, _lib
from ._lib import RpnCfg


def _cfg():
    c = RpnCfg()
    c.outX, c.outY = Constants.nx // 2, Constants.ny // 2
    c.vx, c.vy = Constants.voxelx * 2, Constants.voxely * 2
    for i, a in enumerate(Constants.anchors):
        for j in range(4):
            c.anchors[i][j] = float(a[j])
    return c


def rpnToRegion(labelsClass, labelsRegress, maxBoxes=20, overlapThresh=0.):
    """rpnToRegion(labelsClass (100,200,2), labelsRegress (100,200,14)) -> (boxes (k,7), probs (k,))
    (rpnToRegion.py:113-164; the reference hard-codes maxBoxes=20, overlapThresh=0.).  Inputs may be numpy
    arrays or device tensors (e.g. views of LisecNet's head buffer).  Probability ties pick the larger flat
    index (the reference's np.argsort order among equal keys is unspecified).
    Deviation from the reference, on purpose: nonMaxSuppressionFast deletes the suppressed candidates with
    np.delete(idxs, toDelete) where toDelete holds box INDICES (rpnToRegion.py:66-67) -- a positional delete that
    removes unrelated entries and raises IndexError on numpy >= 1.19 for any realistic map.  Here the boxes found to
    overlap (IoU > overlapThresh) or to lie out of range are the ones suppressed (by value)."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    cfg = _cfg()
    cls = torch.as_tensor(labelsClass, dtype=torch.float32).to(dev)
    reg = torch.as_tensor(labelsRegress, dtype=torch.float32).to(dev)
    cls = cls.reshape(cfg.outX, cfg.outY, -1)
    reg = reg.reshape(cfg.outX, cfg.outY, -1)
    if cls.stride(-1) != 1 or reg.stride(-1) != 1:
        cls, reg = cls.contiguous(), reg.contiguous()
    ws = torch.empty(lib.lisec_rpn_to_region_workspace_bytes(ctypes.byref(cfg), maxBoxes), dtype=torch.uint8, device=dev)
    boxes = torch.zeros((maxBoxes + 1, 7), dtype=torch.float64, device=dev)
    probs = torch.zeros(maxBoxes + 1, dtype=torch.float64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.lisec_rpn_to_region(ctypes.byref(cfg), _lib.ptr(cls), cls.stride(1), _lib.ptr(reg), reg.stride(1),
                                       float(overlapThresh), int(maxBoxes), _lib.ptr(ws), ws.numel(), _lib.ptr(boxes),
                                       _lib.ptr(probs), _lib.ptr(count), _lib.current_stream()))
    k = int(count.item())
    return boxes[:k].cpu().numpy(), probs[:k].cpu().numpy()


def preprocessLabels(data, seed=0, balance=True):
    """preprocessLabels(data (B,7) rows x,y,z,l,w,h,yaw) -> [outClass (100,200,2), outRegress (100,200,14)]
    float64 (serialize_data.py:194-338).  The anchors x boxes IoU sweep runs on the GPU; the region balancing of
    :310-325 uses random.Random(seed) where the reference draws from the unseeded module-level `random`."""
    dev = _lib.require_gpu()
    lib = _lib.load()
    cfg = _cfg()
    data = np.asarray(data, dtype=np.float64).reshape(-1, 7)
    fixed = data.copy()                                    # fixBoxScaling (:181-191)
    fixed[:, [0, 3]] *= cfg.outX / Constants.nx
    fixed[:, [1, 4]] *= cfg.outY / Constants.ny
    B = len(fixed)
    d_fixed = torch.from_numpy(np.ascontiguousarray(fixed)).to(dev) if B else None
    cells = cfg.outX * cfg.outY
    valid = torch.empty(cells * 2, dtype=torch.float64, device=dev)
    overlap = torch.empty(cells * 2, dtype=torch.float64, device=dev)
    outreg = torch.empty(cells * 14, dtype=torch.float64, device=dev)
    ws = torch.empty(lib.lisec_rpn_labels_workspace_bytes(B), dtype=torch.uint8, device=dev)
    _lib.check(lib.lisec_rpn_labels(ctypes.byref(cfg), _lib.ptr(d_fixed), B, float(Constants.iouLowerBound),
                                    float(Constants.iouUpperBound), _lib.ptr(ws), ws.numel(), _lib.ptr(valid),
                                    _lib.ptr(overlap), _lib.ptr(outreg), _lib.current_stream()))
    valid = valid.cpu().numpy().reshape(cfg.outX, cfg.outY, 2)
    overlap = overlap.cpu().numpy().reshape(cfg.outX, cfg.outY, 2)
    outreg = outreg.cpu().numpy().reshape(cfg.outX, cfg.outY, 14)
    if balance:
        _balance(valid, overlap, Constants.maxRegions, seed)
    return [valid + overlap, outreg + np.repeat(overlap, 7, axis=2)]


def _balance(valid, overlap, max_regions, seed):
    """serialize_data.py:310-325: keep <= maxRegions/2 positives and as many negatives as positives."""
    rng = random.Random(seed)
    pos = np.where(np.logical_and(valid == 1, overlap == 1))
    neg = np.where(np.logical_and(valid == 1, overlap == 0))
    pos_count = len(pos[0])
    if pos_count > max_regions / 2:
        locs = rng.sample(range(pos_count), int(pos_count - max_regions / 2))
        valid[pos[0][locs], pos[1][locs], pos[2][locs]] = 0
        pos_count = max_regions / 2
    if len(neg[0]) + pos_count > max_regions:
        locs = rng.sample(range(len(neg[0])), len(neg[0]) - int(pos_count))
        valid[neg[0][locs], neg[1][locs], neg[2][locs]] = 0


def quaternion_yaw(q):
    """pyquaternion's yaw_pitch_roll[0] for a (w,x,y,z) quaternion (serialize_data.py:360-361)."""
    w, x, y, z = (float(v) for v in q)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return math.atan2(2 * (w * z - x * y), 1 - 2 * (y * y + z * z))
