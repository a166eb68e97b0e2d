"""Pin the CPU voxeliser oracle against vectors produced by the reference itself
(tests/golden/make_voxel_goldens.py ran /root/reference/serialize_data.py:97-137)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, LYFT
from oracle import voxel_ref


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "voxel_*_s*.npz"))))
def test_coords_and_counts_bit_exact(name):
    g = _load(name)
    out = voxel_ref.voxelize_ref(g["points"].astype(np.float64), **LYFT)
    assert out["coords"].shape == g["coords"].shape
    assert (out["coords"] == g["coords"].astype(np.int32)).all()
    assert (out["counts"] == g["counts"]).all()
    assert (out["npts"] == g["npts"].astype(np.int32)).all()
    # index 0 is never occupied on any axis (strict lower bounds, model_training.py:118-120)
    assert out["coords"].min() >= 1


@pytest.mark.parametrize("name", ["voxel_small_s0.npz", "voxel_small_s1.npz", "voxel_ring_s0.npz"])
def test_feature_rows_match_reference(name):
    g = _load(name)
    out = voxel_ref.voxelize_ref(g["points"].astype(np.float64), **LYFT)
    keep = np.nonzero(g["counts"] <= 35)[0]
    rows, vid = [], []
    for v in keep:
        r = out["feats"][v, :out["npts"][v]].astype(np.float64)
        o = np.lexsort((r[:, 2], r[:, 1], r[:, 0]))
        rows.append(r[o])
        vid.append(np.full(len(r), v))
    rows, vid = np.concatenate(rows), np.concatenate(vid)
    assert (vid == g["row_voxel"]).all()
    ref32 = g["rows"].astype(np.float32).astype(np.float64)   # the cast Keras applies to its input
    # absolute coordinates are float32-representable -> exact; centroid offsets differ only
    # by the summation order of np.mean over an unseeded permutation -> <= 1 float32 ulp
    assert (rows[:, :3] == ref32[:, :3]).all()
    ulp = np.spacing(np.maximum(np.abs(ref32[:, 3:]), 2.0 ** -20).astype(np.float32)).astype(np.float64)
    assert (np.abs(rows[:, 3:] - ref32[:, 3:]) <= ulp).all()


def test_boundary_known_answers():
    g = _load("voxel_boundary.npz")
    for p, kept, cell in zip(g["points"], g["kept"], g["cell"]):
        out = voxel_ref.voxelize_ref(p[None, :], **LYFT)
        assert len(out["coords"]) == int(kept), p
        if kept:
            assert tuple(out["coords"][0]) == tuple(int(c) for c in cell), p


def test_literal_equals_vectorised():
    rng = np.random.default_rng(7)
    pts = np.stack([rng.uniform(-2.2, 2.2, 900), rng.uniform(-1.1, 1.1, 900), rng.uniform(0, 2.2, 900)], 1)
    pts = pts.astype(np.float32).astype(np.float64)
    small = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=3, maxVoxelX=4, maxVoxelY=4, maxVoxelZ=8)
    idx, val, shape = voxel_ref.voxelize_literal(pts, **small)
    dense_lit = np.zeros(shape)
    for i, v in zip(idx, val):
        dense_lit[i] = v
    out = voxel_ref.voxelize_ref(pts, **small)
    assert out["counts"].max() > 3          # the > sampleSize branch is exercised
    dense = voxel_ref.to_dense(out, shape)
    assert np.array_equal(dense, dense_lit.astype(np.float32))


def test_empty_and_all_filtered():
    out = voxel_ref.voxelize_ref(np.zeros((0, 3)), **LYFT)
    assert out["coords"].shape == (0, 3) and out["feats"].shape == (0, 35, 6)
    out = voxel_ref.voxelize_ref(np.array([[1000.0, 0, 1], [0, 0, -5.0]]), **LYFT)
    assert out["coords"].shape == (0, 3)
