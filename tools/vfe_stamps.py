"""Phase timing inside the VFE stage kernels (diagnostic build path: 100 MHz s_memrealtime stamps of wave 0 of every
workgroup).  Prints, per stage, the median over workgroups of each phase in microseconds and the spread of the starts."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import u20k_cloud  # noqa: E402
from lisec_amd import Constants, _lib  # noqa: E402
from lisec_amd.params import ParamStore  # noqa: E402
from lisec_amd.vfe import VFEStack  # noqa: E402
from lisec_amd.voxelizer import Voxelizer  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                Constants.ny // 2, Constants.nz, device=dev)
vfe = VFEStack(ParamStore(dev), dev)
grid = torch.empty((8, 200, 400, 64), dtype=torch.float32, device=dev)
sample = vox(torch.from_numpy(u20k_cloud(0)).to(dev))
for _ in range(5):
    vfe.forward(sample, True, out=grid)
buf = torch.zeros(2 * 4096 * 8, dtype=torch.int64, device=dev)
lib.lisec_debug_vfe_stamps.argtypes = [ctypes.c_void_p]
_lib.check(lib.lisec_debug_vfe_stamps(buf.data_ptr()))
torch.cuda.synchronize()
vfe.forward(sample, True, out=grid)
torch.cuda.synchronize()
_lib.check(lib.lisec_debug_vfe_stamps(None))
st = buf.cpu().numpy().reshape(2, 4096, 8)
names = ["entry->loads issued", "->prologue done (pre-barrier)", "->barrier", "->first voxel done", "->loop done",
         "->block reduce + atomics issued", "->atomics performed"]
for si, stage in enumerate((2, 3)):
    t = st[si]
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"stage {stage}: {len(t)} workgroups; first start 0, last start {(t[:, 0].max() - t0) / 100:.2f} us, "
          f"last end {(t[:, 7].max() - t0) / 100:.2f} us")
    seq = [0, 1, 2, 3, 4, 5, 6, 7]
    for a, b, n in zip(seq[:-1], seq[1:], names):
        d = (t[:, b] - t[:, a]) / 100.0
        print(f"   {n:36s} median {np.median(d):6.2f} us   max {d.max():6.2f}")
