"""lisec_amd -- MI355X-native hot path of Lisec (voxeliser, VFE, 3D-conv middle, RPN).

Host side: Python mirroring the reference's function-level interface
(model_training.py / Predict.py / Constants.py); compute: hand-written gfx950 HIP
kernels behind the C ABI of include/lisec_hip.h, loaded with ctypes from
lisec_amd/liblisec_hip.so.  There is no CPU fallback: importing the compute modules
without the library raises.
"""
__version__ = "0.1.0"
