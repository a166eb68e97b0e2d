"""The stride-1 convolutions of an RPN block: weight gradients one launch per layer vs one batched launch (alone on the chip)."""
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


for block, (n, hw, ch) in {"rpn3": (5, (25, 50), 256), "rpn2": (5, (50, 100), 128), "rpn1": (3, (100, 200), 128)}.items():
    dims = (1, *hw)
    g = ops.geom(0, dims, dims, (1, 3, 3), (1, 1, 1), (0, 1, 1), ch, ch)
    items = []
    for _ in range(n):
        items.append((g, torch.randn(*dims, ch, device=dev), torch.randn(*dims, ch, device=dev),
                      torch.empty(9, ch, ch, device=dev), torch.randn(4 * ch, device=dev), ops.IN_RELU, False))
    batch = ops.WgradBatch(items)
    ws = torch.zeros(max(batch.workspace_bytes(), ops.wgrad_workspace_bytes(g)), dtype=torch.uint8, device=dev)
    fl = 2.0 * hw[0] * hw[1] * 9 * ch * ch * n
    t_b = timeit(lambda: batch.run(ws))

    def separate():
        for (g_, x, dy, dW, bn, fl_, tr) in items:
            ops.conv_wgrad(g_, x, dy, dW, ws, in_bn=bn, flags=fl_)
    t_s = timeit(separate)
    print(f"{block}: {n} layers, batched {t_b:7.1f} us = {fl / t_b / 1e6 / 157.3:.2f} of peak; one by one {t_s:7.1f} us = "
          f"{fl / t_s / 1e6 / 157.3:.2f}; plan of one layer alone {ops.wgrad_plan(g, flags=ops.IN_RELU)}", flush=True)
