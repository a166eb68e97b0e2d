"""Grid / anchor constants -- same names and values as the reference's Constants.py:1-44."""
import math

# Directory for Lyft dataset (Constants.py:4 hard-codes a Windows path; override via env)
import os as _os
lyft_data_dir = _os.environ.get("LISEC_LYFT_DATA_DIR", "lyft_data")

# size of voxel (Constants.py:7-9)
voxelx = 0.5
voxely = 0.25
voxelz = 0.25

# Number of voxels in space that we care about (Constants.py:12-14): -50..50 m, 0..2 m
nx = int(100 / voxelx)
ny = int(100 / voxely)
nz = int(2 / voxelz)

# anchors (Constants.py:17)
anchors = [[1.6, 3.9, 1.56, 0], [1.6, 3.9, 1.56, math.pi / 2]]

# Limit of points per voxel (Constants.py:20)
maxPoints = 35

# index of points in input tensor (Constants.py:23)
pointIndex = -2

# map of categories (Constants.py:26-36)
catToNum = {
    'car': 0, 'pedestrian': 1, 'animal': 2, 'other_vehicle': 3, 'bus': 4,
    'motorcycle': 5, 'truck': 6, 'emergency_vehicle': 7, 'bicycle': 8,
}

# RPN constants (Constants.py:42-44)
maxRegions = 256
iouLowerBound = 0.45
iouUpperBound = 0.6
