#!/usr/bin/env python3
"""Golden vectors for the pure-numpy label-side helpers, made by RUNNING the reference function.

Run in the build container only:   python tests/golden/make_box_goldens.py

Imports /root/reference/serialize_data.py unmodified, exactly as make_voxel_goldens.py does (the same inert
placeholder modules stand in for the four absent third-party imports; none of them is touched by what runs here),
and calls its fixBoxScaling (serialize_data.py:184-191) on a few shapes, alone and as applied at :217.

Round 4 adds the pure-numpy halves of the box code either side of the network, run the same way (box_geometry_decode.npz):
  * boxToShapely's four vertices, calculateUnion and calculateIntersection's z expression (serialize_data.py:140-169) with a
    Polygon stand-in that only records its vertices and reports an intersection area the generator chose;
  * applyRegrssionNP (rpnToRegion.py:90-113) and the decoded boxInfo / probInfo that rpnToRegion hands to
    nonMaxSuppressionFast (:118-162), with the suppression replaced by a recorder of its arguments.
What stays unpinned: the polygon intersection AREA itself (shapely) and the suppression loop (which the reference cannot
run on a real map, see oracle/boxes_ref.py: nms).  The fixtures hold inputs and outputs only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_voxel_goldens import REF, _install_placeholders  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class _RecordingPolygon:
    """Stand-in for shapely.geometry.Polygon that only RECORDS: the vertex list it was built from, and -- for
    calculateIntersection -- an `intersection` whose area is a number the generator chose, so that the reference's own
    arithmetic AROUND the polygon library (corner convention, z extent, union) is what produces the golden values."""
    area_of_intersection = 1.0

    def __init__(self, pts):
        self.pts = [tuple(float(v) for v in p) for p in pts]

    def intersection(self, other):
        import types as _t
        return _t.SimpleNamespace(area=_RecordingPolygon.area_of_intersection)


def _install_box_placeholders():
    """What `import rpnToRegion` / `import serialize_data` need at module scope beyond make_voxel_goldens' placeholders:
    matplotlib (rpnToRegion.py:1,12-14 picks the TkAgg backend), shapely.ops, and model_training (whose import pulls in
    Keras); none of them is touched by the functions run here."""
    import types

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    mpl = mod("matplotlib", use=lambda *a, **k: None)
    mpl.pyplot = mod("matplotlib.pyplot")
    mpl.patches = mod("matplotlib.patches")
    mod("shapely.ops", cascaded_union=lambda *a, **k: None)
    sys.modules["shapely.geometry"].Polygon = _RecordingPolygon
    mod("model_training", combine_lidar_data=lambda *a, **k: None)


def box_geometry_goldens(ref, rng):
    """boxToShapely's four vertices, calculateUnion, and calculateIntersection's z expression (serialize_data.py:140-169)
    on seeded boxes (x, y, z, l, w, h, yaw), with the polygon area fixed by the stand-in."""
    out = {}
    n = 24
    boxes = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-1, 3, n), rng.uniform(0.5, 6, n),
                      rng.uniform(0.5, 6, n), rng.uniform(0.5, 3, n), rng.uniform(-4, 4, n)], 1)
    boxes[0, 6] = 0.0
    boxes[1, 6] = np.pi / 2
    boxes[2, 6] = -np.pi / 2
    boxes[4, 2], boxes[5, 2] = -10.0, 10.0              # far apart in z: the reference's z extent is not clamped -> negative
    out["geom_boxes"] = boxes
    out["geom_corners"] = np.array([ref.boxToShapely(list(b)).pts for b in boxes], dtype=np.float64)      # (n, 4, 2)
    areas = rng.uniform(0.0, 5.0, n // 2)
    inter, union, iou = [], [], []
    for k in range(n // 2):
        b1, b2 = list(boxes[2 * k]), list(boxes[2 * k + 1])
        _RecordingPolygon.area_of_intersection = float(areas[k])
        i_ = ref.calculateIntersection(b1, b2)
        inter.append(i_)
        union.append(ref.calculateUnion(b1, b2, i_))
        iou.append(ref.calculateIoU(b1, b2))
    out["geom_pair_area"] = areas
    out["geom_pair_intersection"] = np.array(inter, dtype=np.float64)
    out["geom_pair_union"] = np.array(union, dtype=np.float64)
    out["geom_pair_iou"] = np.array(iou, dtype=np.float64)
    return out


def decode_goldens(rng):
    """rpnToRegion.py:90-162 run as written: applyRegrssionNP on seeded anchor / regression maps, and rpnToRegion on seeded
    (100,200,2) / (100,200,14) maps with nonMaxSuppressionFast replaced by a recorder of its arguments -- the decoded
    boxInfo / probInfo are what the reference hands to the suppression (:162)."""
    import rpnToRegion as rref
    out = {}
    X = rng.normal(0, 3, (7, 6, 9))
    X[3:6] = np.abs(X[3:6]) + 0.5
    t = rng.normal(0, 0.4, (7, 6, 9))
    out["regr_X"], out["regr_t"] = X, t
    out["regr_out"] = np.asarray(rref.applyRegrssionNP(X, t), dtype=np.float64)
    cls = rng.uniform(0, 1, (100, 200, 2))
    reg = rng.normal(0, 0.3, (100, 200, 14))
    seen = {}

    def recorder(boxInfo, probInfo, **kw):
        seen["boxInfo"], seen["probInfo"], seen["kw"] = np.array(boxInfo, dtype=np.float64), np.array(probInfo, dtype=np.float64), kw
        return [], []
    keep = rref.nonMaxSuppressionFast
    rref.nonMaxSuppressionFast = recorder
    try:
        rref.rpnToRegion(cls, reg)
    finally:
        rref.nonMaxSuppressionFast = keep
    out["decode_cls"], out["decode_reg"] = cls.astype(np.float32), reg.astype(np.float32)
    # the maps are stored as float32 (what the network emits, Predict.py:38-40); the reference ran on exactly those values
    seen.clear()
    rref.nonMaxSuppressionFast = recorder
    try:
        rref.rpnToRegion(out["decode_cls"].astype(np.float64), out["decode_reg"].astype(np.float64))
    finally:
        rref.nonMaxSuppressionFast = keep
    out["decode_boxInfo"], out["decode_probInfo"] = seen["boxInfo"], seen["probInfo"]
    out["decode_nms_args"] = np.array([seen["kw"]["maxBoxes"], seen["kw"]["overlapThresh"]], dtype=np.float64)
    return out


def main():
    _install_placeholders()
    _install_box_placeholders()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import serialize_data as ref
    rng2 = np.random.default_rng(7)
    geo = box_geometry_goldens(ref, rng2)
    geo.update(decode_goldens(rng2))
    np.savez_compressed(os.path.join(OUT, "box_geometry_decode.npz"), **geo)
    print("wrote box_geometry_decode.npz", {k: v.shape for k, v in geo.items()})
    cases = [((1, 7), 100, 200, 200, 400), ((5, 7), 100, 200, 200, 400), ((3, 7), 50, 25, 200, 400),
             ((12, 7), 100, 200, 100, 200)]
    out = {}
    for i, (shape, nx_, ny_, ox, oy) in enumerate(cases):
        out[f"args{i}"] = np.array(list(shape) + [nx_, ny_, ox, oy], dtype=np.int64)
        out[f"mult{i}"] = np.asarray(ref.fixBoxScaling(shape, nx_, ny_, ox, oy), dtype=np.float64)
    rng = np.random.default_rng(0)
    data = rng.normal(0, 10, (5, 7))
    out["data"] = data
    out["fixed"] = data * ref.fixBoxScaling(data.shape, 100, 200, 200, 400)       # the call of serialize_data.py:217
    np.savez(os.path.join(OUT, "box_fixscaling.npz"), **out)
    print("wrote box_fixscaling.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
