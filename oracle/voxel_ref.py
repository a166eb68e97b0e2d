"""CPU oracle for the Lisec voxeliser -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (lisec_amd/) never does: it fails loudly when the
HIP library is missing.

Restates, in numpy, the algorithm of the reference functions

    get_voxel           /root/reference/model_training.py:103-107
    VFE_preprocessing   /root/reference/model_training.py:112-152
    (byte-identical copies: serialize_data.py:88-137)

Pinned against golden vectors produced by running the reference function itself
(tests/golden/make_voxel_goldens.py -> tests/golden/voxel_*.npz):
  * occupied voxel coordinates and min(count, sampleSize): bit-exact;
  * per-voxel feature rows: equal as a SET of rows (the reference permutes the
    slot order with an unseeded np.random.choice, model_training.py:132) for
    every voxel with count <= sampleSize.
What is NOT pinned by the reference: which points survive when a voxel holds
more than sampleSize points, and the slot order.  The oracle's deterministic
policy is "ascending original point index, first sampleSize kept".

Output convention (what the HIP voxeliser must reproduce bit for bit):
  coords  int32 (V,3)  rows (z, x, y) exactly as the reference emits its COO
                        index prefix (model_training.py:148), sorted by the
                        linear key (z*NX + x)*NY + y
  counts  int32 (V,)    raw number of in-range points of the voxel
  npts    int32 (V,)    min(count, sampleSize)
  feats   float32 (V, sampleSize, 6)  [x, y, z, x-cx, y-cy, z-cz], float64
                        arithmetic (the reference works on float64 arrays) rounded
                        once to float32 (the Keras input dtype); zero padded
"""
from math import floor

import numpy as np


def get_voxel(point, xSize, ySize, zSize):
    """model_training.py:103-107 -- floor of a float64 divide, per axis."""
    return (floor(point[0] / xSize), floor(point[1] / ySize), floor(point[2] / zSize))


def voxel_keys(points, xSize, ySize, zSize, maxVoxelX, maxVoxelY, maxVoxelZ):
    """Vectorised pass 1 (model_training.py:115-122).

    Returns (valid mask, fx, fy, fz) with fx = kx + maxVoxelX etc. ("fixedKey").
    The range test is STRICT on both sides, so kx == -maxVoxelX, ky == -maxVoxelY
    and kz == 0 are dropped (model_training.py:118-120).
    """
    p = np.asarray(points, dtype=np.float64)
    kx = np.floor(p[:, 0] / xSize)
    ky = np.floor(p[:, 1] / ySize)
    kz = np.floor(p[:, 2] / zSize)
    valid = (
        (-maxVoxelX < kx) & (kx < maxVoxelX)
        & (-maxVoxelY < ky) & (ky < maxVoxelY)
        & (0 < kz) & (kz < maxVoxelZ)
    )
    # NaN compares false everywhere -> invalid, like the reference (floor(nan) raises
    # there; we simply drop such points)
    valid &= np.isfinite(kx) & np.isfinite(ky) & np.isfinite(kz)
    fx = np.where(valid, kx + maxVoxelX, 0).astype(np.int64)
    fy = np.where(valid, ky + maxVoxelY, 0).astype(np.int64)
    fz = np.where(valid, kz, 0).astype(np.int64)
    return valid, fx, fy, fz


def voxelize_ref(points, xSize, ySize, zSize, sampleSize, maxVoxelX, maxVoxelY, maxVoxelZ):
    """Deterministic restatement of VFE_preprocessing (model_training.py:112-152).

    Returns dict(coords, counts, npts, feats, point_index) -- see module docstring.
    point_index int32 (V, sampleSize): original index of the point in each slot, -1
    for pad rows.
    """
    p = np.asarray(points, dtype=np.float64)
    nxg, nyg = 2 * maxVoxelX, 2 * maxVoxelY
    valid, fx, fy, fz = voxel_keys(p, xSize, ySize, zSize, maxVoxelX, maxVoxelY, maxVoxelZ)
    idx = np.nonzero(valid)[0]
    lin = (fz[idx] * nxg + fx[idx]) * nyg + fy[idx]
    order = np.argsort(lin, kind="stable")  # stable: ascending point index inside a voxel
    lin_s, idx_s = lin[order], idx[order]
    ukeys, start, counts = np.unique(lin_s, return_index=True, return_counts=True)
    V = len(ukeys)
    coords = np.empty((V, 3), dtype=np.int32)
    coords[:, 0] = ukeys // (nxg * nyg)
    coords[:, 1] = (ukeys // nyg) % nxg
    coords[:, 2] = ukeys % nyg
    npts = np.minimum(counts, sampleSize).astype(np.int32)
    feats = np.zeros((V, sampleSize, 6), dtype=np.float32)
    pidx = np.full((V, sampleSize), -1, dtype=np.int32)
    for v in range(V):
        s = int(npts[v])
        sel = idx_s[start[v]:start[v] + s]           # lowest `s` point indices
        cur = p[sel]                                  # model_training.py:134
        # np.mean(axis=0) on a C-contiguous (s,3) float64 array adds the rows in
        # order and divides once (model_training.py:135); written out so that the C
        # restatement and the HIP kernel can follow the same order of operations.
        acc = np.zeros(3, dtype=np.float64)
        for r in range(s):
            acc = acc + cur[r]
        centroid = acc / np.float64(s)
        block = np.hstack((cur, cur - centroid))      # model_training.py:137-140
        feats[v, :s, :] = block.astype(np.float32)    # Keras casts its input to float32
        pidx[v, :s] = sel
    return dict(coords=coords, counts=counts.astype(np.int32), npts=npts, feats=feats,
                point_index=pidx)


def voxelize_literal(points, xSize, ySize, zSize, sampleSize, maxVoxelX, maxVoxelY, maxVoxelZ):
    """Loop-for-loop restatement (small inputs only) used to cross-check voxelize_ref.

    Follows model_training.py:113-152 statement by statement, with the unseeded
    np.random.choice (line 132) replaced by "first s indices of the bucket" (buckets
    are filled in ascending point order, line 124).
    Returns (indices list of 5-tuples, values list, dense_shape) like the SparseTensor
    the reference builds at lines 151-152.
    """
    clustered = {}
    for idx, point in enumerate(points):
        key = get_voxel(point, xSize, ySize, zSize)
        if -maxVoxelX < key[0] < maxVoxelX and -maxVoxelY < key[1] < maxVoxelY \
                and 0 < key[2] < maxVoxelZ:
            fixed = (key[0] + maxVoxelX, key[1] + maxVoxelY, key[2])
            clustered.setdefault(fixed, []).append(idx)
    appended = {}
    for voxel, bucket in clustered.items():
        s = sampleSize if len(bucket) > sampleSize else len(bucket)
        cur = np.asarray(points, dtype=np.float64)[bucket[:s]]
        centroid = np.mean(cur, axis=0)
        concat = np.hstack((cur, cur[:, 0:1] - centroid[0], cur[:, 1:2] - centroid[1],
                            cur[:, 2:3] - centroid[2]))
        appended[voxel] = np.vstack((concat, np.zeros((sampleSize - s, 6))))
    indices, values = [], []
    for voxel, buf in appended.items():
        for i in range(len(buf)):
            for j in range(len(buf[i])):
                indices.append((voxel[2],) + voxel[:2] + (i, j))
                values.append(buf[i][j])
    return indices, values, [maxVoxelZ, maxVoxelX * 2, maxVoxelY * 2, sampleSize, 6]


def to_dense(vox, dense_shape):
    """tf.sparse.to_dense equivalent (model_training.py:279) for a voxelize_ref result."""
    dense = np.zeros(dense_shape, dtype=np.float32)
    c = vox["coords"]
    dense[c[:, 0], c[:, 1], c[:, 2]] = vox["feats"]
    return dense
