"""HIP voxeliser (through the C ABI) vs the CPU oracle and the reference's golden vectors."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, LYFT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vox():
    from lisec_amd.voxelizer import Voxelizer
    return Voxelizer(**LYFT)


def _check_against_oracle(got, ref):
    assert np.array_equal(got["coords"], ref["coords"])
    assert np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["npts"], ref["npts"])
    assert np.array_equal(got["point_index"], ref["point_index"])
    assert np.array_equal(got["feats"], ref["feats"])          # bit-exact, fp64 math rounded once


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "voxel_*_s*.npz"))))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_golden_clouds(vox, name, dtype):
    from oracle import voxel_ref
    g = np.load(os.path.join(GOLDEN, name))
    pts = g["points"].astype(dtype)
    s = vox(pts)
    got = s.to_host()
    # reference golden: indices and min(count, 35) bit-exact
    assert np.array_equal(got["coords"], g["coords"].astype(np.int32))
    assert np.array_equal(got["counts"], g["counts"])
    assert np.array_equal(got["npts"], g["npts"].astype(np.int32))
    assert s.host_info()["max_count"] == int(g["counts"].max())
    _check_against_oracle(got, voxel_ref.voxelize_ref(pts.astype(np.float64), **LYFT))


def test_boundary_points(vox):
    g = np.load(os.path.join(GOLDEN, "voxel_boundary.npz"))
    for p, kept, cell in zip(g["points"], g["kept"], g["cell"]):
        got = vox(p[None, :]).to_host()
        assert len(got["coords"]) == int(kept), p
        if kept:
            assert tuple(got["coords"][0]) == tuple(int(c) for c in cell), p


def test_empty_nan_and_strided_input(vox):
    from oracle import voxel_ref
    assert vox(np.zeros((0, 3), np.float32)).host_info()["V"] == 0
    pts = np.array([[np.nan, 0, 1], [1e30, 0, 1], [0.3, 0.3, 0.6], [-np.inf, 0, 1]], np.float64)
    assert vox(pts).host_info()["V"] == 1
    # lidar .bin layout: 5 floats per point (model_training.py:87-90)
    rng = np.random.default_rng(5)
    raw = rng.uniform(-20, 20, (4000, 5)).astype(np.float32)
    raw[:, 2] = rng.uniform(0, 2, 4000)
    _check_against_oracle(vox(raw).to_host(), voxel_ref.voxelize_ref(raw[:, :3].astype(np.float64), **LYFT))


def test_crowded_voxels_keep_lowest_indices(vox):
    """> 64 and > 35 points in one voxel: deterministic 'lowest 35 point indices' policy."""
    from oracle import voxel_ref
    rng = np.random.default_rng(11)
    a = np.stack([rng.uniform(1.0, 1.5, 500), rng.uniform(1.0, 1.25, 500), rng.uniform(0.5, 0.75, 500)], 1)
    b = np.stack([rng.uniform(-3.0, -2.5, 50), rng.uniform(2.0, 2.25, 50), rng.uniform(1.0, 1.25, 50)], 1)
    c = np.stack([rng.uniform(-30, 30, 3000), rng.uniform(-30, 30, 3000), rng.uniform(0.3, 1.9, 3000)], 1)
    pts = np.concatenate([a, b, c])
    rng.shuffle(pts)
    got = vox(pts).to_host()
    ref = voxel_ref.voxelize_ref(pts, **LYFT)
    assert ref["counts"].max() > 64
    _check_against_oracle(got, ref)


def test_r200k_full_size_properties(vox):
    """Lyft-size cloud (n ~ 200 000, model_training.py:116): size-independent checks."""
    import torch
    rng = np.random.default_rng(3)
    n = 200_000
    az = rng.uniform(0, 2 * np.pi, n)
    r = 2.0 + 68.0 * rng.uniform(0, 1, n) ** 2
    pts = np.stack([r * np.cos(az), r * np.sin(az), rng.uniform(-0.2, 2.2, n)], 1).astype(np.float32)
    s = vox(pts)
    got = s.to_host()
    hi = s.host_info()
    assert got["counts"].sum() == hi["valid"]
    assert (got["npts"] == np.minimum(got["counts"], 35)).all()
    lin = (got["coords"][:, 0] * 200 + got["coords"][:, 1]) * 400 + got["coords"][:, 2]
    assert (np.diff(lin) > 0).all()                                    # sorted, unique cells
    cv = s.cell_voxel.cpu().numpy()
    assert (cv[lin] == np.arange(hi["V"])).all() and (cv >= 0).sum() == hi["V"]
    # every kept row belongs to its voxel and centroid offsets sum to ~0
    rows, rs = got["rows"], got["row_start"]
    vid = np.repeat(np.arange(hi["V"]), got["npts"])
    assert (np.floor(rows[:, 0].astype(np.float64) / 0.5) + 100 == got["coords"][vid, 1]).all()
    assert (np.floor(rows[:, 1].astype(np.float64) / 0.25) + 200 == got["coords"][vid, 2]).all()
    assert (np.floor(rows[:, 2].astype(np.float64) / 0.25) == got["coords"][vid, 0]).all()
    sums = np.zeros((hi["V"], 3))
    np.add.at(sums, vid, rows[:, 3:].astype(np.float64))
    assert np.abs(sums).max() < 1e-4
    # idempotence: same cloud again -> identical output
    again = vox(pts).to_host()
    assert all(np.array_equal(got[k], again[k]) for k in ("coords", "counts", "rows", "row_point"))
    torch.cuda.synchronize()
