"""The reference's label-side script under its own name (serialize_data.py): the functions a Lisec user calls to build
the training targets, now backed by the GPU kernels of lisec_amd.boxes / lisec_amd.model_training.

    rotate_points, combine_lidar_data, get_voxel, VFE_preprocessing      serialize_data.py:56-137  (same code path as
                                                                          model_training.py's copies of them)
    fixBoxScaling, preprocessLabels                                       serialize_data.py:184-338
    imageToRPN, saveLabelsForSample                                       serialize_data.py:341-409

The reference keeps the Lyft dataset in a module global `level5Data` that its __main__ block fills; here the functions
take it as an optional last argument and fall back to the module global of the same name.
"""
import os

import numpy as np

from . import Constants
from .boxes import preprocessLabels, quaternion_yaw                                   # noqa: F401
from .model_training import VFE_preprocessing, get_voxel, rotate_points              # noqa: F401
from .model_training import combine_lidar_data as _combine_lidar_data

level5Data = None            # serialize_data.py:36-40 builds a LyftDataset here


def _dataset(ds):
    ds = ds if ds is not None else level5Data
    if ds is None:
        raise RuntimeError("set lisec_amd.serialize_data.level5Data (or pass the dataset) first")
    return ds


def combine_lidar_data(sample, dataDir, dataset=None):
    """combine_lidar_data(sample, dataDir) (serialize_data.py:64-85)."""
    return _combine_lidar_data(sample, dataDir, _dataset(dataset))


def fixBoxScaling(dataSize, newX, newY, origX, origY):
    """fixBoxScaling(data.shape, newX, newY, origX, origY) (serialize_data.py:184-191): takes the SHAPE (rows, 7) of
    the label table and returns the multiplier matrix of that shape -- columns 0 and 3 (x, length) newX/origX, columns
    1 and 4 (y, width) newY/origY, everything else 1 -- which the caller applies as `data * fixBoxScaling(data.shape,
    ...)` (:217).  preprocessLabels applies the same scaling itself."""
    out = np.ones(tuple(dataSize), dtype=np.float64)
    if out.ndim != 2 or out.shape[1] < 5:
        raise ValueError("fixBoxScaling expects the shape (rows, >= 5) of a label table")
    out[:, 0] = newX / origX
    out[:, 1] = newY / origY
    out[:, 3] = newX / origX
    out[:, 4] = newY / origY
    return out


def imageToRPN(sample, dataset=None, seed=0):
    """imageToRPN(sample) (serialize_data.py:341-381): the cars of one sample within +-50 m, moved from global to
    ego coordinates (translate, then the INVERSE ego rotation), rows [x, y, z, *size, yaw] -> preprocessLabels."""
    ds = _dataset(dataset)
    labels = []
    sd = ds.get('sample_data', sample['data']['LIDAR_TOP'])
    ego = ds.get('ego_pose', sd['ego_pose_token'])
    for token in sample['anns']:
        ann = ds.get('sample_annotation', token)
        t = np.array(ann['translation'], dtype=np.float64).reshape(1, -1) - np.array(ego['translation'])
        t = rotate_points(t, np.array(ego['rotation']), True)
        row = [t[0, 0], t[0, 1], t[0, 2]] + list(ann['size']) + [quaternion_yaw(ann['rotation'])]
        instance = ds.get('instance', ann['instance_token'])
        category = ds.get('category', instance['category_token'])['name']
        if Constants.catToNum[category] == 0 and -50 <= row[0] <= 50 and -50 <= row[1] <= 50:
            labels.append(row)
    outClass, outRegress = preprocessLabels(np.array(labels, dtype=np.float64).reshape(-1, 7), seed=seed)
    return outClass, outRegress


def saveLabelsForSample(samples, outPath, dataset=None):
    """saveLabelsForSample(samples, outPath) (serialize_data.py:384-408): labelsClass.npy (n,100,200,2),
    regressClass.npy (n,100,200,14) and the two shape files, float64 -- what train() loads from 'labels3'."""
    classMap, regressMap = [], []
    for i, s in enumerate(samples):
        print('doing sample', str(i))
        outClass, outRegress = imageToRPN(s, dataset, seed=i)
        print('sample ' + str(i) + ' finished')
        classMap.append(outClass)
        regressMap.append(outRegress)
    classMap, regressMap = np.stack(classMap), np.stack(regressMap)
    os.makedirs(outPath, exist_ok=True)
    np.save(os.path.join(outPath, 'labelsClass.npy'), classMap)
    np.save(os.path.join(outPath, 'regressClass.npy'), regressMap)
    np.save(os.path.join(outPath, 'labelsShape.npy'), np.array(classMap.shape))
    np.save(os.path.join(outPath, 'regressShape.npy'), np.array(regressMap.shape))
    return classMap, regressMap
