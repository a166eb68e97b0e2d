"""Fixed per-tile cost of k_igemm: the same 320 000 output rows (64 -> 64 channels) with 1, 3, 9 and 27 taps.
time = a + b * taps: `a` is prologue + epilogue + launch, `b` one K slab of 64 for every tile (GPU box only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.bench_conv import run

if __name__ == "__main__":
    ts = []
    for name, k, p in (("1 tap", (1, 1, 1), (0, 0, 0)), ("3 taps (kw)", (1, 1, 3), (0, 0, 1)),
                       ("9 taps (kh,kw)", (1, 3, 3), (0, 1, 1)), ("27 taps", (3, 3, 3), (1, 1, 1))):
        run(name, 0, (4, 200, 400), (4, 200, 400), k, (1, 1, 1), p, 64, 64, iters=20)
