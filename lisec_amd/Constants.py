"""Grid, anchor and RPN constants under the names the reference's drivers read (Constants.py:1-44 there):
``Constants.voxelx``, ``Constants.nx``, ``Constants.maxPoints``, ``Constants.anchors`` ...

Everything is derived from three primitives -- the metric extent of the scene, the voxel edge lengths and the
anchor box -- so that a different grid is one edit.  The defaults reproduce the reference's Lyft setup:
100 m x 100 m x 2 m, voxels of 0.5 x 0.25 x 0.25 m -> a (nz, nx, ny) = (8, 200, 400) grid, 35 points per voxel.
"""
import math
import os

# where the Lyft Level-5 dataset lives (the reference hard-codes a Windows path; here: environment variable)
lyft_data_dir = os.environ.get("LISEC_LYFT_DATA_DIR", "lyft_data")

_SCENE_METRES = (100.0, 100.0, 2.0)          # x, y in [-50, 50) m; z in [0, 2) m
_VOXEL_METRES = (0.5, 0.25, 0.25)
voxelx, voxely, voxelz = _VOXEL_METRES
nx, ny, nz = (int(extent / edge) for extent, edge in zip(_SCENE_METRES, _VOXEL_METRES))

maxPoints = 35            # rows kept per voxel (T)
pointIndex = -2           # axis of the point dimension in the (.., T, C) tensors

_ANCHOR_LWH = (1.6, 3.9, 1.56)               # one car-sized box, at yaw 0 and at yaw 90 degrees
anchors = [list(_ANCHOR_LWH) + [yaw] for yaw in (0, math.pi / 2)]

catToNum = {name: i for i, name in enumerate(
    ("car", "pedestrian", "animal", "other_vehicle", "bus", "motorcycle", "truck", "emergency_vehicle", "bicycle"))}

# RPN target generation
maxRegions = 256          # valid anchors kept per sample (half positive at most)
iouLowerBound = 0.45      # below: negative; between: ignored
iouUpperBound = 0.6       # at or above: positive
