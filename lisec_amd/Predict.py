"""Drop-in counterpart of the reference's Predict.py: predictMain(samples, outPath, level5Data, model)."""
import os
import time

import numpy as np

from . import Constants
from .model_training import (MaxPoolingVFELayer, RepeatLayer, VFE_preprocessing, combine_lidar_data,  # noqa: F401
                             load_model, sparse)


def predictMain(samples, outPath, level5Data, model, dataDir=None):
    """For every sample: assemble the lidar sweep, voxelise, run the network in inference mode and save
    outPath/sample{i}_label.npy (1,100,200,2) and outPath/sample{i}_regress.npy (1,100,200,14), float32
    (Predict.py:9-40; the reference joins the path with a literal backslash, i.e. Windows only)."""
    dataDir = dataDir if dataDir is not None else Constants.lyft_data_dir
    os.makedirs(outPath, exist_ok=True)
    for i in range(len(samples)):
        sampleLidarPoints = combine_lidar_data(samples[i], dataDir, level5Data)
        startTime = time.time()
        trainVFEPoints = VFE_preprocessing(sampleLidarPoints, Constants.voxelx, Constants.voxely, Constants.voxelz,
                                           Constants.maxPoints, Constants.nx // 2, Constants.ny // 2, Constants.nz)
        trainVFEPoints = sparse.reshape(trainVFEPoints, (1,) + trainVFEPoints.shape)     # Predict.py:29
        print(time.time() - startTime)
        print('finished ' + str(i))
        prob, regress = model.predict(trainVFEPoints)                                     # Predict.py:38
        np.save(os.path.join(outPath, 'sample' + str(i) + '_label.npy'), prob)
        np.save(os.path.join(outPath, 'sample' + str(i) + '_regress.npy'), regress)


if __name__ == '__main__':
    # python -m lisec_amd.Predict [model.h5] [outPath]      (Predict.py:43-59)
    import sys
    from .model_training import MaxPoolingVFELayer, RepeatLayer, _lyft_dataset, load_model
    level5Data = _lyft_dataset()
    model = load_model(sys.argv[1] if len(sys.argv) > 1 else os.path.join('fixedTheta', '15SampleEpoch0_fixed.h5'),
                       custom_objects={'RepeatLayer': RepeatLayer, 'MaxPoolingVFELayer': MaxPoolingVFELayer})
    samples = [level5Data.get('sample', scene['first_sample_token']) for scene in level5Data.scene]
    print('Testing on ' + str(len(samples)))
    predictMain(samples, sys.argv[2] if len(sys.argv) > 2 else 'fixedTheta', level5Data, model)
