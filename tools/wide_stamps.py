"""Phase timing inside k_igemm_wide on the rpn1.conv1 shape (100 MHz stamps of every workgroup) for several K slicings."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lisec_amd import _lib, ops

dev = "cuda"
lib = _lib.load()
lib.lisec_debug_igemm_stamps.argtypes = [ctypes.c_void_p]
dims, cin, cout = (1, 100, 200), 128, 128
g = ops.geom(0, dims, dims, (1, 3, 3), (1, 1, 1), (0, 1, 1), cin, cout)
x = torch.randn(*dims, cin, device=dev)
w = torch.randn(9, cin, cout, device=dev) * 0.03
wp = ops.pack_weights(w, 9, cin, cout, cin * cout, cout, 1)
out = torch.empty(*dims, cout, device=dev)
bn = torch.randn(4 * cin, device=dev)
for ms in (3,):
    _lib.set_tuning(max_splitk=ms)
    run = lambda: ops.conv_forward(g, x, wp, out, in_bn=bn, flags=ops.IN_RELU)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    _lib.check(lib.lisec_debug_igemm_stamps(buf.data_ptr()))
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    _lib.check(lib.lisec_debug_igemm_stamps(None))
    t = buf.cpu().numpy().reshape(8192, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    plan = ops.conv_plan(g, in_bn=True, flags=ops.IN_RELU)
    print(f"max_splitk {ms}: {us:.1f} us; plan {plan['kernel']} k_slices {plan['k_slices']} workgroups {plan['workgroups']}; {len(t)} stamped; "
          f"starts 0 .. {(t[:, 0].max() - t0) / 100:.1f} us, last end {(t[:, 4].max() - t0) / 100:.1f} us")
    d = lambda a, b: np.median((t[:, b] - t[:, a]) / 100.0)
    loop = (t[:, 3] - t[:, 2]) / 100.0
    ns = np.maximum(t[:, 5], 1)
    cu = ((t[:, 6] >> 32) & 0xf) * 256 + ((t[:, 6] >> 8) & 0xff)
    ids, counts = np.unique(cu, return_counts=True)
    print(f"   setup {d(0, 1):5.2f}  first tile {d(1, 2):5.2f}  loop {np.median(loop):6.2f} ({np.median(loop / ns):.2f} us/step over {np.median(ns):.0f} steps)  "
          f"after loop {d(3, 4):5.2f}  whole {d(0, 4):6.2f}; {len(ids)} CUs, workgroups per CU: "
          + ", ".join(f"{k}: {int((counts == k).sum())}" for k in sorted(set(counts))))
    if plan["k_slices"] > 1:
        arr = (t[:, 7] - t[:, 3]) / 100.0                  # slab store + ticket (+ the last arriver's slab sum)
        fin = (t[:, 4] - t[:, 7]) / 100.0                  # the last arriver's epilogue (others: ~0)
        last = fin > 0.5
        print(f"   slices: store + ticket median {np.median(arr[~last]):.2f} us; last arrivers ({int(last.sum())}): arrive + slab sum median "
              f"{np.median(arr[last]):.2f} p90 {np.percentile(arr[last], 90):.2f}, epilogue median {np.median(fin[last]):.2f} p90 "
              f"{np.percentile(fin[last], 90):.2f} max {fin[last].max():.2f}; loop end: p10 {np.percentile((t[:, 3] - t0) / 100.0, 10):.1f} "
              f"p50 {np.percentile((t[:, 3] - t0) / 100.0, 50):.1f} p90 {np.percentile((t[:, 3] - t0) / 100.0, 90):.1f} max {((t[:, 3] - t0) / 100.0).max():.1f}")
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 4], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0])
    print(f"   mean workgroups alive {np.sum(alive[:-1] * dur) / max(dur.sum(), 1):.0f}, peak {int(alive.max())}")
_lib.set_tuning(max_splitk=12)
