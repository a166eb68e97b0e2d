"""The lidar-ingest oracle (oracle/ingest_ref.py, reference model_training.py:65-98) against hand-computed known
answers, and the product's host-side numpy path (lisec_amd.model_training.rotate_points / combine_lidar_data, a
closed-form rotation matrix) against that oracle (Hamilton sandwich product): two independent derivations."""
import os

import numpy as np
import pytest

from oracle import ingest_ref


@pytest.mark.parametrize("q,inverse,p,want", ingest_ref.KNOWN_ANSWERS)
def test_oracle_known_answers(q, inverse, p, want):
    got = ingest_ref.rotate_points(np.array([p]), q, inverse)
    assert got.shape == (1, 3) and np.allclose(got[0], want, rtol=0, atol=1e-15 * 8)


def test_oracle_is_a_rotation():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(0, 1, 4) * rng.uniform(0.1, 10)             # any non-zero quaternion (normalised inside)
        p = rng.normal(0, 30, (50, 3))
        r = ingest_ref.rotate_points(p, q)
        assert np.allclose(np.linalg.norm(r, axis=1), np.linalg.norm(p, axis=1), rtol=1e-13)      # lengths kept
        assert np.allclose(ingest_ref.rotate_points(r, q, inverse=True), p, rtol=1e-12, atol=1e-12)
        e = ingest_ref.rotate_points(np.eye(3), q)                                                    # rows = R^T
        assert np.linalg.det(e) > 0.999999                                                           # proper, no mirror
        # composition: rotating by q2 after q1 is the Hamilton product q2*q1
        q2 = rng.normal(0, 1, 4)
        both = ingest_ref.rotate_points(ingest_ref.rotate_points(p, q), q2)
        prod = ingest_ref._hamilton(q2 / np.linalg.norm(q2), q / np.linalg.norm(q))
        assert np.allclose(both, ingest_ref.rotate_points(p, prod), rtol=1e-11, atol=1e-11)


class _Level5:
    def __init__(self, root, rng, sensors):
        self.t = {"sample_data": {}, "calibrated_sensor": {}}
        self.sample = {"data": {}}
        os.makedirs(os.path.join(root, "lidar"), exist_ok=True)
        for j, s in enumerate(sensors):
            raw = rng.normal(0, 20, (500 + 10 * j, 5)).astype(np.float32)
            fn = f"lidar/{s}.bin"
            raw.tofile(os.path.join(root, fn))
            self.t["sample_data"][f"sd{j}"] = {"filename": fn.replace("/", "\\") if j == 1 else fn,
                                               "calibrated_sensor_token": f"cs{j}"}
            self.t["calibrated_sensor"][f"cs{j}"] = {"rotation": list(rng.normal(0, 1, 4) * (1 + j)),     # not unit
                                                     "translation": list(rng.normal(0, 2, 3))}
            self.sample["data"][s] = f"sd{j}"

    def get(self, table, token):
        return self.t[table][token]


def test_product_host_path_matches_oracle(tmp_path):
    """lisec_amd's numpy rotate_points / combine_lidar_data (no GPU needed) == the oracle, incl. non-unit quaternions,
    a missing sensor (model_training.py:75-79) and a Windows-style file name (:85)."""
    from lisec_amd import model_training as mt
    rng = np.random.default_rng(1)
    for q, inverse, p, want in ingest_ref.KNOWN_ANSWERS:
        assert np.allclose(mt.rotate_points(np.array([p]), q, inverse)[0], want, atol=1e-14)
    for sensors in (ingest_ref.SENSOR_TYPES, ("LIDAR_TOP", "LIDAR_FRONT_LEFT")):
        root = tmp_path / ("s%d" % len(sensors))
        l5 = _Level5(str(root), rng, sensors)
        want = ingest_ref.combine_lidar_data(l5.sample, str(root), l5)
        got = mt.combine_lidar_data(l5.sample, str(root), l5)
        assert got.dtype == np.float64 and got.shape == want.shape == (sum(500 + 10 * j for j in range(len(sensors))), 3)
        assert np.allclose(got, want, rtol=1e-13, atol=1e-12)
