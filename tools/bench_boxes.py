"""Timing of the two "next" rows around the hot path: label generation (serialize_data.py:194-338) and RPN decode +
rotated NMS (rpnToRegion.py:18-164) on the GPU, with the CPU oracle (the reference's algorithm with the shapely polygon
clipping restated) beside it on a reduced case.  GPU box only."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lisec_amd import boxes
from oracle import boxes_ref as B


def scene(rng, n):
    d = np.zeros((n, 7))
    d[:, 0] = rng.uniform(-48, 48, n); d[:, 1] = rng.uniform(-48, 48, n); d[:, 2] = rng.uniform(0.5, 1.5, n)
    d[:, 3] = rng.uniform(3.5, 5.2, n); d[:, 4] = rng.uniform(6.5, 9.0, n); d[:, 5] = rng.uniform(1.3, 1.8, n)
    d[:, 6] = rng.choice([0.0, np.pi / 2, 0.1, -0.2, 1.4], n)
    return d


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (10, 40):
        data = scene(rng, n)
        t = timed(lambda: boxes.preprocessLabels(data, seed=1), 5)
        print(f"preprocessLabels  {n:3d} boxes x 40 000 anchors   GPU {t * 1e3:8.2f} ms per sample (incl. host copies)")
    data = scene(rng, 10)
    t0 = time.perf_counter()
    B.preprocess_labels(data, seed=1)
    print(f"preprocess_labels  10 boxes   CPU oracle (python, circumradius pre-test)  {time.perf_counter() - t0:8.2f} s")
    cls = rng.uniform(0, 1, (1, 100, 200, 2)).astype(np.float32)
    reg = rng.normal(0, 0.1, (1, 100, 200, 14)).astype(np.float32)
    for mb in (20, 100):
        t = timed(lambda: boxes.rpnToRegion(cls, reg, maxBoxes=mb), 5)
        print(f"rpnToRegion  40 000 candidates, maxBoxes {mb:3d}   GPU {t * 1e3:8.2f} ms")
    t0 = time.perf_counter()
    bi = B.decode_boxes(reg[0].astype(np.float64))
    pi = np.concatenate([cls[0, :, :, a].reshape(-1) for a in range(2)]).astype(np.float64)
    B.nms(bi[:4000], pi[:4000], maxBoxes=20)
    print(f"nms   4 000 candidates (a tenth), maxBoxes 20   CPU oracle  {time.perf_counter() - t0:8.2f} s")
