"""Winograd form against the direct form of the same contraction, alone on the chip, at the Lyft geometries (us per call,
median of `reps` after warm-up; algorithmic TFLOP/s = the DIRECT form's 2 * M * taps * Cin * Cout / time for both).
usage: python tools/wino_bench.py [reps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from lisec_amd import ops  # noqa: E402

H, W = 200, 400
CASES = [
    ("mid2 forward", 0, (4, H, W), (2, H, W), 3, 1, 0, 64, 64, False),
    ("mid3 forward", 0, (2, H, W), (1, H, W), 3, 2, 1, 64, 64, False),
    ("rpn1.conv1 forward", 0, (1, 100, 200), (1, 100, 200), 1, 1, 0, 128, 128, True),
    ("mid2 data gradient", 1, (2, H, W), (4, H, W), 3, 1, 0, 64, 64, False),
    ("mid3 data gradient", 1, (1, H, W), (2, H, W), 3, 2, 1, 64, 64, False),
    ("rpn1.conv1 data gradient", 1, (1, 100, 200), (1, 100, 200), 1, 1, 0, 128, 128, False),
    ("rpn2.conv1 forward", 0, (1, 50, 100), (1, 50, 100), 1, 1, 0, 128, 128, True),
]


def live_taps(mode, ind, outd, KD, sd, pd):
    n = 0
    for o in range(outd[0]):
        for kd in range(KD):
            if mode == 0:
                s = o * sd - pd + kd
                n += 0 <= s < ind[0]
            else:
                t = o + pd - kd
                n += t >= 0 and t % sd == 0 and t // sd < ind[0]
    return n * 9


def timeit(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    for name, mode, ind, outd, KD, sd, pd, cin, cout, xf in CASES:
        k, s, p = (KD, 3, 3), (sd, 1, 1), (pd, 1, 1)
        g = ops.geom(mode, ind, outd, k, s, p, cin, cout)
        x = torch.from_numpy(rng.normal(0, 1, (*ind, cin)).astype(np.float32)).to(dev)
        Wt = torch.from_numpy((rng.normal(0, 1, (KD * 9, cin, cout)) / np.sqrt(KD * 9 * cin)).astype(np.float32)).to(dev)
        bn = torch.from_numpy(np.concatenate([rng.uniform(0.5, 1.5, cin), rng.normal(0, 0.3, cin), np.zeros(cin),
                                              np.ones(cin)]).astype(np.float32)).to(dev) if xf else None
        flags = ops.IN_RELU if xf else 0
        wp = ops.pack_weights(Wt, KD * 9, cin, cout, cin * cout, cout, 1)
        wu = ops.pack_weights_winograd(Wt, KD, cin, cout, cin * cout, cout, 1, flip=(mode == 1))
        out = torch.empty(*outd, cout, device=dev)
        out2 = torch.empty(*outd, cout, device=dev)
        t_d = timeit(lambda: ops.conv_forward(g, x, wp, out, in_bn=bn, flags=flags), reps)
        t_w = timeit(lambda: ops.conv_forward_winograd(g, x, wu, out2, in_bn=bn, flags=flags), reps)
        if len(sys.argv) > 2:        # timing-only variants (wrong results): 0x1000 every patch load reads element 0, 0x2000 one U image
            for dbg in (0x3000, 0x4000, 0x8000, 0xb000):
                out3 = torch.empty(*outd, cout, device=dev)
                t = timeit(lambda: ops.conv_forward_winograd(g, x, wu, out3, in_bn=bn, flags=flags | dbg), reps)
                print(f"    debug {dbg:#x}: {t:7.1f} us")
        err = float((out - out2).norm() / out.norm())
        M1 = outd[1] * outd[2]
        gf = 2.0 * M1 * live_taps(mode, ind, outd, KD, sd, pd) * cin * cout / 1e9
        print(f"{name:28s} direct {t_d:7.1f} us ({gf / t_d * 1e3:6.1f} TF/s)   winograd {t_w:7.1f} us ({gf / t_w * 1e3:6.1f} TF/s algorithmic)"
              f"   x{t_d / t_w:.2f}   |direct - winograd| / |direct| = {err:.1e}", flush=True)


def wgrad_main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda")
    rng = np.random.default_rng(1)
    for name, ind, outd, KD, sd, pd in [("mid2 weight gradient", (4, H, W), (2, H, W), 3, 1, 0),
                                        ("mid3 weight gradient", (2, H, W), (1, H, W), 3, 2, 1)]:
        g = ops.geom(0, ind, outd, (KD, 3, 3), (sd, 1, 1), (pd, 1, 1), 64, 64)
        x = torch.from_numpy(rng.normal(0, 1, (*ind, 64)).astype(np.float32)).to(dev)
        dy = torch.from_numpy(rng.normal(0, 1, (*outd, 64)).astype(np.float32)).to(dev)
        dW, dW2 = torch.empty(KD * 9, 64, 64, device=dev), torch.empty(KD * 9, 64, 64, device=dev)
        ws = torch.zeros(ops.wgrad_workspace_bytes(g), dtype=torch.uint8, device=dev)
        ws2 = torch.empty(ops.wgrad_winograd_workspace_bytes(g), dtype=torch.uint8, device=dev)
        t_d = timeit(lambda: ops.conv_wgrad(g, x, dy, dW, ws), reps)
        t_w = timeit(lambda: ops.conv_wgrad_winograd(g, x, dy, dW2, ws2), reps)
        err = float((dW - dW2).norm() / dW.norm())
        gf = 2.0 * outd[1] * outd[2] * live_taps(0, ind, outd, KD, sd, pd) * 64 * 64 / 1e9
        print(f"{name:28s} ring   {t_d:7.1f} us ({gf / t_d * 1e3:6.1f} TF/s)   winograd {t_w:7.1f} us ({gf / t_w * 1e3:6.1f} TF/s algorithmic)"
              f"   x{t_d / t_w:.2f}   |ring - winograd| / |ring| = {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
    wgrad_main()
