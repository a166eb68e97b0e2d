"""Which CUs does a stream made with hipExtStreamCreateWithCUMask use?  Launches a stamped contraction on masked streams and
lists the (XCC, CU) pairs its workgroups ran on.  usage: python tools/cumask_probe.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lisec_amd import _lib, ops

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
dev = torch.device("cuda")
lib = _lib.load()


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def where(stream, label):
    x = torch.randn(1, 100, 200, 128, device=dev)
    w = torch.randn(9, 128, 128, device=dev) * 0.05
    wp = ops.pack_weights(w, 9, 128, 128, 128 * 128, 128, 1)
    out = torch.empty(1, 100, 200, 128, device=dev)
    g = ops.geom(0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128)
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(buf.data_ptr()))
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.conv_forward(g, x, wp, out)
        e0.record()
        for _ in range(5):
            ops.conv_forward(g, x, wp, out)
        e1.record()
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    xcc = (t[:, 6] >> 32) & 0xf
    cu = (t[:, 6] >> 8) & 0xff
    pairs = sorted(set(zip(xcc.tolist(), cu.tolist())))
    per_xcc = {x_: sum(1 for p in pairs if p[0] == x_) for x_ in sorted(set(xcc.tolist()))}
    print(f"{label}: {len(pairs)} CUs used, per XCC {per_xcc}, {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call")


where(torch.cuda.current_stream(), "default stream")
where(masked_stream(range(256)), "mask all 256 bits")
where(masked_stream(range(128)), "mask bits 0..127")
where(masked_stream(range(128, 256)), "mask bits 128..255")
where(masked_stream(range(0, 256, 2)), "mask even bits")
where(masked_stream([b for b in range(256) if (b % 8) < 5]), "mask bits with b%8 < 5")
where(masked_stream(range(32)), "mask bits 0..31")
