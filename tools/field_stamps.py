"""Phase timing inside k_field_taps / k_field_combine (100 MHz s_memrealtime stamps of thread 0 of every workgroup)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from lisec_amd import Constants, _lib, ops
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer

dev = torch.device("cuda")
lib = _lib.load()
lib.lisec_debug_field_stamps.argtypes = [ctypes.c_void_p]
net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints)
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2,
                Constants.ny // 2, Constants.nz)
c = net.layers[0]["conv"]
names = {0: ["entry->V known", "->coords checked", "->A,W in LDS", "->MFMAs of tap 0", "->stores issued + barrier", "->end"],
         1: ["entry->cells+constants in LDS", "->masks", "->positions written", "->sums issued", "->sink finished"]}
for cloud_name in sys.argv[1:] or ["u20k"]:
    cloud = bench.u20k_cloud(0) if cloud_name == "u20k" else bench.r200k_cloud(0)
    sample = vox(torch.from_numpy(cloud).to(dev))
    for _ in range(3):
        net.forward(sample, training=True)
    buf = torch.zeros(2 * 8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_field_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    ops.conv_field_forward(c.g, net.vfe.saved_field("vout"), net.vfe.saved_field("delta"), sample, net.packed[c.name],
                           net.act["mid1.y"], net.field_ws, bias=net.params.view(c.bias), sink=net._fwd_sink(c))
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_field_stamps(None))
    st = buf.cpu().numpy().reshape(2, 8192, 8)
    t00 = st[0][st[0][:, 0] > 0][:, 0].min()
    for k in (0, 1):
        t = st[k]
        t = t[t[:, 0] > 0]
        last = len(names[k])
        full = t[t[:, last] > 0]
        print(f"{cloud_name} kernel {k}: {len(t)} workgroups started, {len(full)} ran to the end; first start "
              f"{(t[:, 0].min() - t00) / 100:.2f} us, last start {(t[:, 0].max() - t00) / 100:.2f} us, last end "
              f"{(full[:, last].max() - t00) / 100:.2f} us")
        for i, n in enumerate(names[k]):
            d = (full[:, i + 1] - full[:, i]) / 100.0
            print(f"   {n:34s} median {np.median(d):6.2f} us   max {d.max():6.2f}")
