"""lisec_amd -- MI355X-native hot path of Lisec (voxeliser, VFE, 3D-conv middle, RPN).

Host side: Python mirroring the reference's function-level interface
(model_training.py / Predict.py / Constants.py); compute: hand-written gfx950 HIP
kernels behind the C ABI of include/lisec_hip.h, loaded with ctypes from
lisec_amd/liblisec_hip.so.  There is no CPU fallback: importing the compute modules
without the library raises.
"""
__version__ = "0.1.0"

import os as _os

# The HIP runtime hands streams to GPU_MAX_HW_QUEUES hardware queues (default 4).  A data-parallel step has five streams of
# ours (main, second, exchange) and RCCL's own: with four queues they share, and a wait on one stream holds back the launches
# of the stream it shares a queue with (one rank through RCCL: 5.2 ms per step with 4 queues, 4.34 with 8; the one-GPU line
# is unchanged).  Read when the runtime initialises, so it is set at import, before anything touches the GPU; an explicit
# setting of the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
