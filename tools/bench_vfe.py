"""Event timing of the voxeliser and of the whole VFE forward call (training), U20k and R200k sweeps.
LISEC_VFE_SHAPE selects the launch shape of the stage kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import event_time_ms, r200k_cloud, u20k_cloud  # noqa: E402
from lisec_amd import Constants  # noqa: E402
from lisec_amd.params import ParamStore  # noqa: E402
from lisec_amd.vfe import VFEStack  # noqa: E402
from lisec_amd.voxelizer import Voxelizer  # noqa: E402

dev = torch.device("cuda")
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                Constants.ny // 2, Constants.nz, device=dev)
vfe = VFEStack(ParamStore(dev), dev)
grid = torch.empty((8, 200, 400, 64), dtype=torch.float32, device=dev)
for name, cloud in (("u20k", u20k_cloud(0)), ("r200k", r200k_cloud(0))):
    pts = torch.from_numpy(cloud).to(dev)
    sample = vox(pts)
    ms_vox = event_time_ms(lambda: vox(pts), 50)
    ms_vfe = event_time_ms(lambda: vfe.forward(sample, True, out=grid), 50)
    ms_inf = event_time_ms(lambda: vfe.forward(sample, False, out=grid), 50)
    hi = sample.host_info()
    byt = 12.0 * len(cloud) + 24.0 * hi["rows"] + 4.0 * 64 * 8 * 200 * 400
    print(f"shape {os.environ.get('LISEC_VFE_SHAPE', 'default')} {name}: voxelise {ms_vox * 1e3:.1f} us, vfe forward (training) "
          f"{ms_vfe * 1e3:.1f} us = {byt / ms_vfe / 1e6:.0f} GB/s = {byt / ms_vfe / 1e6 / 8000:.3f} of 8 TB/s, inference "
          f"{ms_inf * 1e3:.1f} us; V {hi['V']} rows {hi['rows']}")
