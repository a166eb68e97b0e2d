"""The 128-channel w-halo layers of the first RPN block alone: 128 x 128 tiles (k_igemm_wide) against the 128 x 32 /
128 x 64 plans (lisec_tuning.wide_tile = 0), forward with a statistics sink and data gradient with a backward sink."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lisec_amd import _lib, ops

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, mode, dims, cin, cout in [("rpn1.conv1 forward", 0, (1, 100, 200), 128, 128), ("rpn1.conv1 data gradient", 1, (1, 100, 200), 128, 128),
                                    ("up1 (256 channels) forward", 1, (1, 100, 200), 128, 256)]:
    M = dims[1] * dims[2]
    g = ops.geom(mode, dims, dims, (1, 3, 3), (1, 1, 1), (0, 1, 1), cin, cout)
    x = torch.randn(*dims, cin, device=dev)
    w = torch.randn(9, cin, cout, device=dev) * 0.03
    wp = ops.pack_weights(w, 9, cin, cout, cin * cout, cout, 1)
    out = torch.empty(*dims, cout, device=dev)
    bn = torch.randn(4 * cin, device=dev)
    gamma, beta, st = torch.ones(cout, device=dev), torch.zeros(cout, device=dev), torch.zeros(4 * cout, device=dev)
    y = torch.randn(*dims, cout, device=dev)
    fl = 2.0 * M * 9 * cin * cout
    for wide in (1, 0):
        _lib.set_tuning(wide_tile=wide)
        if mode == 0:
            sink = ops.BnSink(cout, M, dev, gamma=gamma, beta=beta, bnstate=st)
            kw = dict(in_bn=bn, flags=ops.IN_RELU, sink=sink)
        elif cout == 128:
            sink = ops.BnSink(cout, M, dev, dgamma=torch.zeros(cout, device=dev), dbeta=torch.zeros(cout, device=dev))
            kw = dict(bwd=(y, torch.randn(4 * cout, device=dev), True), sink=sink)
        else:
            kw = dict(in_bn=bn, flags=ops.IN_RELU)
        plan = ops.conv_plan(g, **{k: (True if k == "in_bn" else v) for k, v in kw.items()})
        us = timeit(lambda: ops.conv_forward(g, x, wp, out, **kw))
        print(f"{name}: wide_tile={wide}: {us:6.1f} us = {fl / us / 1e6 / 157.3:.2f} of peak; plan {plan['kernel']} cols {plan['cols']} "
              f"k_slices {plan['k_slices']} workgroups {plan['workgroups']}", flush=True)
_lib.set_tuning(wide_tile=1)
