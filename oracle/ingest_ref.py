"""TEST INFRASTRUCTURE -- CPU oracle of the lidar ingest step (reference model_training.py:65-98).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product
(lisec_amd/) never does.

    rotate_points(points, rotation, inverse)      model_training.py:65-69
    combine_lidar_data(sample, dataDir, level5)   model_training.py:73-98

The reference rotates with pyquaternion (`Quaternion(rotation).rotation_matrix`, a third-party
package that is absent from /root/reference and not installed here; version unpinned by the
reference, README.md:5-10).  Its published semantics, restated: the (w, x, y, z) quaternion is
NORMALISED first, then acts by the Hamilton sandwich product  v' = q (0, v) q*  (right-handed,
active rotation; `.inverse` is the conjugate of the unit quaternion).  This file evaluates exactly
that sandwich product with explicit Hamilton products -- deliberately NOT the closed-form 3x3 matrix
the product code uses (lisec_amd.model_training._quaternion_matrix), so that the two derivations
check each other.

Pinning: pyquaternion cannot run here, so the convention is pinned by hand-computed known answers
(KNOWN_ANSWERS below: identity, 90 degrees about each axis, 180 degrees, a non-unit quaternion, the
inverse) -- PARITY UNPINNED beyond those and the algebraic identities tests/test_oracle_ingest.py checks.
"""
import os

import numpy as np

SENSOR_TYPES = ('LIDAR_TOP', 'LIDAR_FRONT_RIGHT', 'LIDAR_FRONT_LEFT')      # model_training.py:74

_S = np.sqrt(0.5)
# (quaternion wxyz, inverse flag, input point, expected output) -- every row worked out by hand:
# a rotation by +90 degrees about z takes x -> y, y -> -x; about x: y -> z, z -> -y; about y: z -> x, x -> -z.
KNOWN_ANSWERS = [
    ((1.0, 0.0, 0.0, 0.0), False, (1.0, 2.0, 3.0), (1.0, 2.0, 3.0)),             # identity
    ((_S, 0.0, 0.0, _S), False, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),               # +90 about z
    ((_S, 0.0, 0.0, _S), False, (0.0, 1.0, 0.0), (-1.0, 0.0, 0.0)),
    ((_S, 0.0, 0.0, _S), False, (1.0, 2.0, 3.0), (-2.0, 1.0, 3.0)),
    ((_S, _S, 0.0, 0.0), False, (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)),               # +90 about x
    ((_S, _S, 0.0, 0.0), False, (1.0, 2.0, 3.0), (1.0, -3.0, 2.0)),
    ((_S, 0.0, _S, 0.0), False, (0.0, 0.0, 1.0), (1.0, 0.0, 0.0)),               # +90 about y
    ((_S, 0.0, _S, 0.0), False, (1.0, 2.0, 3.0), (3.0, 2.0, -1.0)),
    ((0.0, 0.0, 0.0, 1.0), False, (1.0, 2.0, 3.0), (-1.0, -2.0, 3.0)),           # 180 about z
    ((2.0, 0.0, 0.0, 2.0), False, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),             # NON-UNIT: normalised first
    ((-_S, 0.0, 0.0, -_S), False, (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),             # q and -q: same rotation
    ((_S, 0.0, 0.0, _S), True, (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),                # inverse: -90 about z
    ((_S, 0.0, 0.0, _S), True, (1.0, 2.0, 3.0), (2.0, -1.0, 3.0)),
    ((0.5, 0.5, 0.5, 0.5), False, (1.0, 2.0, 3.0), (3.0, 1.0, 2.0)),             # 120 about (1,1,1): x->y->z->x
]


def _hamilton(a, b):
    """Hamilton product of (..., 4) quaternion arrays, (w, x, y, z) order."""
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], -1)


def rotate_points(points, rotation, inverse=False):
    """model_training.py:65-69: `np.dot(Quaternion(rotation)[.inverse].rotation_matrix, points.T).T`, float64."""
    q = np.asarray(rotation, dtype=np.float64).reshape(4)
    q = q / np.sqrt((q * q).sum())                       # pyquaternion normalises before rotating
    if inverse:
        q = q * np.array([1.0, -1.0, -1.0, -1.0])        # inverse of a unit quaternion = conjugate
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    v = np.concatenate([np.zeros((len(p), 1)), p], 1)    # pure quaternions (0, v)
    qc = q * np.array([1.0, -1.0, -1.0, -1.0])
    out = _hamilton(_hamilton(np.broadcast_to(q, v.shape), v), np.broadcast_to(qc, v.shape))
    return out[:, 1:]


def combine_lidar_data(sample, dataDir, level5Data):
    """model_training.py:73-98: for every lidar sensor the sample has (fixed order, missing ones skipped :75-79),
    read the flat float32 .bin (:87), keep x,y,z of each 5-float row (:90), rotate by the sensor's quaternion (:93),
    add its translation (:94), concatenate (:96).  float64 (n, 3)."""
    out = []
    for sensor_type in SENSOR_TYPES:
        if sensor_type not in sample['data']:
            continue
        frame = level5Data.get('sample_data', sample['data'][sensor_type])
        sensor = level5Data.get('calibrated_sensor', frame['calibrated_sensor_token'])
        path = os.path.join(dataDir, *frame['filename'].replace('\\', '/').split('/'))   # :85 joins with '\\'
        raw = np.fromfile(path, dtype=np.float32).reshape(-1, 5)[:, :3]
        out.append(rotate_points(raw, sensor['rotation']) + np.asarray(sensor['translation'], dtype=np.float64))
    return np.concatenate(out)
