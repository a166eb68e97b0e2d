"""CPU oracle for the box geometry around the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/ (and smoke/bench checker legs) may import this module.

Restates, in plain Python/numpy float64,

    boxToShapely / calculateIntersection / calculateUnion / calculateIoU   serialize_data.py:138-178
    preprocessLabels (anchor grid x ground-truth boxes -> RPN targets)      serialize_data.py:194-338
    applyRegrssion / rpnToRegion / nonMaxSuppressionFast                    rpnToRegion.py:18-164

PINNED BY THE REFERENCE ITSELF (round 4, tests/golden/make_box_goldens.py runs serialize_data.py / rpnToRegion.py
unmodified under recording stand-ins; tests/golden/box_geometry_decode.npz): box_corners (boxToShapely's vertices),
intersection_volume / union_volume / the IoU quotient around a GIVEN polygon area, apply_regression, and decode_boxes +
the probability order (what rpnToRegion hands to nonMaxSuppressionFast).
STILL UNPINNED: the polygon intersection AREA, which the reference delegates to shapely (third party, absent
here and unpinned); it is restated as Sutherland-Hodgman clipping of two convex quadrilaterals + the shoelace
formula, and cross-checked against an independent Monte-Carlo area estimate in tests/test_oracle_boxes.py.
The reference's unseeded random.sample balancing (serialize_data.py:310-325) and np.argsort tie order
(rpnToRegion.py:40) are not reproducible; the deterministic stand-ins are documented at each function.
Quirks kept on purpose: the z overlap uses the FULL height as half-extent and is not clamped
(serialize_data.py:144-146), so an "intersection" and an IoU can be negative; the union uses l*w*h.
"""
import math
import random

import numpy as np


def box_corners(box):
    """boxToShapely (serialize_data.py:149-162): corners [topRight, botRight, botLeft, topLeft]."""
    theta, length, width = box[6], box[3], box[4]
    rx, ry = box[0] + math.cos(theta) * (width / 2), box[1] - math.sin(theta) * (width / 2)
    lx, ly = box[0] - math.cos(theta) * (width / 2), box[1] + math.sin(theta) * (width / 2)
    sx, sy = math.sin(theta) * (length / 2), math.cos(theta) * (length / 2)
    return [(rx + sx, ry + sy), (rx - sx, ry - sy), (lx - sx, ly - sy), (lx + sx, ly + sy)]


def _signed_area(poly):
    a = 0.0
    for i in range(len(poly)):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % len(poly)]
        a += x0 * y1 - x1 * y0
    return 0.5 * a


def convex_intersection_area(p, q):
    """Area of the intersection of two convex polygons (Sutherland-Hodgman, clip p by the edges of q)."""
    if _signed_area(p) < 0:
        p = p[::-1]
    if _signed_area(q) < 0:
        q = q[::-1]
    out = list(p)
    for i in range(len(q)):
        ax, ay = q[i]
        bx, by = q[(i + 1) % len(q)]
        ex, ey = bx - ax, by - ay
        inp, out = out, []
        if not inp:
            break
        for j in range(len(inp)):
            cx, cy = inp[j]
            dx, dy = inp[(j + 1) % len(inp)]
            sc = ex * (cy - ay) - ey * (cx - ax)        # >= 0: inside (left of the edge)
            sd = ex * (dy - ay) - ey * (dx - ax)
            if sc >= 0:
                out.append((cx, cy))
            if (sc >= 0) != (sd >= 0):
                t = sc / (sc - sd)
                out.append((cx + t * (dx - cx), cy + t * (dy - cy)))
    if len(out) < 3:
        return 0.0
    return abs(_signed_area(out))


def intersection_volume(box1, box2, area):
    """calculateIntersection (serialize_data.py:140-147) around the polygon area: the z extent uses the FULL height as
    half-extent and is not clamped.  Pinned by a reference run with a chosen area (tests/golden/box_geometry_decode.npz)."""
    botZ = max(box1[2] - box1[5], box2[2] - box2[5])
    topZ = min(box1[2] + box1[5], box2[2] + box2[5])
    return (topZ - botZ) * area


def union_volume(box1, box2, intersect):
    """calculateUnion (serialize_data.py:165-168)."""
    return box1[3] * box1[4] * box1[5] + box2[3] * box2[4] * box2[5] - intersect


def calculate_iou(box1, box2, area=None):
    """calculateIoU (serialize_data.py:170-178) with calculateIntersection / calculateUnion (:138-167).  area: the polygon
    intersection area when it is given from outside (parity checks); None = this file's own clipping."""
    if area is None:
        area = convex_intersection_area(box_corners(box1), box_corners(box2))
    intersect = intersection_volume(box1, box2, area)
    return intersect / union_volume(box1, box2, intersect)


def apply_regression(X, regress):
    """applyRegrssionNP (rpnToRegion.py:90-113): X anchors (7, ...), regress (7, ...) -> decoded (7, ...)."""
    X, t = np.asarray(X, dtype=np.float64), np.asarray(regress, dtype=np.float64)
    return np.stack([t[0] * X[3] + X[0], t[1] * X[4] + X[1], t[2] * X[5] + X[2], np.exp(t[3]) * X[3], np.exp(t[4]) * X[4],
                     np.exp(t[5]) * X[5], t[6] + X[6]])


# ---------------------------------------------------------------------------------------------------
def preprocess_labels(data, nx=200, ny=400, voxelx=0.5, voxely=0.25, anchors=None, iou_lo=0.45, iou_hi=0.6,
                      max_regions=256, seed=0, balance=True):
    """preprocessLabels (serialize_data.py:194-338).  data: (B,7) rows x,y,z,l,w,h,yaw.
    Returns [outClass (outX,outY,A), outRegress (outX,outY,7A)].

    Balancing (:310-325) draws with an unseeded `random.sample` in the reference; here random.Random(seed)
    is used over the same index lists (np.where order), so WHICH regions are switched off is not comparable
    with a reference run, only how many."""
    if anchors is None:
        anchors = [[1.6, 3.9, 1.56, 0], [1.6, 3.9, 1.56, math.pi / 2]]
    data = np.asarray(data, dtype=np.float64).reshape(-1, 7)
    outX, outY = nx // 2, ny // 2
    vx, vy = voxelx * 2, voxely * 2
    A = len(anchors)
    outRegress = np.zeros((outX, outY, A * 7))
    valid = np.zeros((outX, outY, A))
    overlap = np.zeros((outX, outY, A))
    B = len(data)
    bestIou = np.zeros(B)
    bestAnchor = -np.ones((B, 3), dtype=int)
    count = np.zeros(B)
    bestReg = np.zeros((B, 7))
    fixed = data.copy()                                   # fixBoxScaling (:181-191): x,l by outX/nx; y,w by outY/ny
    fixed[:, 0] *= outX / nx
    fixed[:, 3] *= outX / nx
    fixed[:, 1] *= outY / ny
    fixed[:, 4] *= outY / ny
    centerZ = 1.0
    # a rectangle pair further apart than the sum of the circumradii has polygon area 0 -> IoU exactly 0
    rad = [0.5 * math.hypot(b[3], b[4]) for b in fixed]
    for i in range(A):
        arad = 0.5 * math.hypot(anchors[i][0], anchors[i][1])
        for xV in range(int(-outX / 2), int(outX / 2)):
            cX = vx * xV + vx / 2
            if cX - anchors[i][0] / 2 < vx * int(-outX / 2) or cX + anchors[i][0] / 2 > vx * int(outX / 2):
                continue
            for yV in range(int(-outY / 2), int(outY / 2)):
                cY = vy * yV + vy / 2
                if cY - anchors[i][1] / 2 < vy * int(-outY / 2) or cY + anchors[i][1] / 2 > vy * int(outY / 2):
                    continue
                boxType, bestLoc, bestRegression = 'neg', 0, (0,) * 7
                anchorBox = [cX, cY, centerZ] + list(anchors[i])
                for b in range(B):
                    fb = fixed[b]
                    if math.hypot(fb[0] - cX, fb[1] - cY) > rad[b] + arad:
                        iou = 0.0
                    else:
                        iou = calculate_iou(anchorBox, fb)
                    if iou > bestIou[b] or iou >= iou_hi or iou > iou_lo:
                        reg = ((fb[0] - cX) / anchorBox[3], (fb[1] - cY) / anchorBox[4], (fb[2] - centerZ) / anchorBox[5],
                               math.log(fb[3] / anchorBox[3]), math.log(fb[4] / anchorBox[4]),
                               math.log(fb[5] / anchorBox[5]), fb[6] - anchorBox[6])
                    if iou > bestIou[b]:
                        bestIou[b] = iou
                        bestAnchor[b] = (xV, yV, i)
                        bestReg[b] = reg
                    if iou >= iou_hi:
                        boxType = 'pos'
                        count[b] += 1
                        if iou > bestLoc:
                            bestLoc, bestRegression = iou, reg
                    if iou_lo < iou <= iou_hi and boxType != 'pos':
                        boxType = 'neutral'
                if boxType == 'neg':
                    valid[xV, yV, i] = 1                   # negative indices wrap, as in the reference (:284-294)
                elif boxType == 'pos':
                    valid[xV, yV, i] = 1
                    overlap[xV, yV, i] = 1
                    outRegress[xV, yV, i * 7:i * 7 + 7] = bestRegression
    for b in range(B):                                     # every box gets at least one positive anchor (:297-307)
        if count[b] == 0 and bestIou[b] != 0:
            a = bestAnchor[b]
            valid[a[0], a[1], a[2]] = 1
            overlap[a[0], a[1], a[2]] = 1
            outRegress[a[0], a[1], a[2] * 7:a[2] * 7 + 7] = bestReg[b]
    if balance:
        balance_regions(valid, overlap, max_regions, seed)
    return [valid + overlap, outRegress + np.repeat(overlap, 7, axis=2)]


def balance_regions(valid, overlap, max_regions=256, seed=0):
    """:310-325 with random.Random(seed) instead of the unseeded module-level random."""
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


# ---------------------------------------------------------------------------------------------------
def decode_boxes(labelsRegress, nx=200, ny=400, voxelx=0.5, voxely=0.25, anchors=None):
    """The anchor grid + applyRegrssion part of rpnToRegion (rpnToRegion.py:75-150).
    Returns boxInfo (A*outX*outY, 7), rows ordered anchor-major like the reference's reshape (:146-148)."""
    if anchors is None:
        anchors = [[1.6, 3.9, 1.56, 0], [1.6, 3.9, 1.56, math.pi / 2]]
    outX, outY = nx // 2, ny // 2
    vx, vy = voxelx * 2, voxely * 2
    reg = np.asarray(labelsRegress, dtype=np.float64)
    boxes = np.zeros((len(anchors), outX, outY, 7))
    X = np.arange(outX)[:, None] * vx + vx / 2
    Y = np.arange(outY)[None, :] * vy + vy / 2
    for i, a in enumerate(anchors):
        t = reg[:, :, i * 7:i * 7 + 7]
        boxes[i, :, :, 0] = t[:, :, 0] * a[0] + X
        boxes[i, :, :, 1] = t[:, :, 1] * a[1] + Y
        boxes[i, :, :, 2] = t[:, :, 2] * a[2] + 1.0
        boxes[i, :, :, 3] = np.exp(t[:, :, 3]) * a[0]
        boxes[i, :, :, 4] = np.exp(t[:, :, 4]) * a[1]
        boxes[i, :, :, 5] = np.exp(t[:, :, 5]) * a[2]
        boxes[i, :, :, 6] = t[:, :, 6] + a[3]
    return boxes.reshape(-1, 7)


def nms(boxInfo, probInfo, overlapThresh=0.0, maxBoxes=20, anchor=(1.6, 3.9)):
    """nonMaxSuppressionFast (rpnToRegion.py:18-72).  Ties in probability: the larger flat index is taken first
    (the reference's np.argsort order among equal keys is unspecified).  Returns picked indices.

    INTENDED semantics, one deliberate deviation: the reference collects the suppressed BOX indices in `toDelete`
    and then calls np.delete(idxs, toDelete) (:66-67), which removes by POSITION in the shrinking `idxs` list, not by
    value -- it drops unrelated candidates and raises IndexError (numpy >= 1.19) as soon as a box index exceeds the
    list length, i.e. on every real 40 000-candidate map.  This restatement (and boxes.hip, which is checked against
    it) suppresses the boxes that were actually found to overlap / lie out of range: by value, via an alive[] mask.
    The positional delete is NOT reproduced."""
    alive = np.ones(len(probInfo), dtype=bool)
    pick = []
    while alive.any():
        cand = np.nonzero(alive)[0]
        best = cand[np.lexsort((cand, probInfo[cand]))[-1]]
        pick.append(int(best))
        alive[best] = False
        last = boxInfo[best]
        for j in np.nonzero(alive)[0]:
            x, y = boxInfo[j, 0], boxInfo[j, 1]
            if x - anchor[0] < 0 or x + anchor[0] > 100 or y - anchor[1] < 0 or y + anchor[1] > 100:
                alive[j] = False
            else:
                # rectangles further apart than the sum of their circumradii: polygon area 0 -> IoU exactly 0
                far = math.hypot(x - last[0], y - last[1]) > 0.5 * (math.hypot(last[3], last[4]) +
                                                                    math.hypot(boxInfo[j, 3], boxInfo[j, 4]))
                iou = 0.0 if far else calculate_iou(list(last), list(boxInfo[j]))
                if iou > overlapThresh:
                    alive[j] = False
        if len(pick) > maxBoxes:
            break
    return pick
