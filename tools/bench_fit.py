"""Throughput of the reference-named surface: Model.fit(batch_size=1) on voxelised sweeps (GPU box only)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import synthetic_targets, u20k_cloud
from lisec_amd import Constants
from lisec_amd import model_training as mt

if __name__ == "__main__":
    n = 4
    pts = [u20k_cloud(i).astype(np.float64) for i in range(n)]
    samples = [mt.VFE_preprocessing(p, Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints,
                                    Constants.nx // 2, Constants.ny // 2, Constants.nz) for p in pts]
    tg = [synthetic_targets(i, Constants.nx // 2, Constants.ny // 2) for i in range(n)]
    ycls = np.stack([t[0] for t in tg]).astype(np.float64)
    yreg = np.stack([t[1] for t in tg]).astype(np.float64)
    model = mt.createModel(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints)
    model.compile(optimizer=mt.optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True), loss=['mse', 'mse'])
    model.fit(x=samples, y=[ycls, yreg], batch_size=1, verbose=0, epochs=1, steps_per_epoch=20)
    torch.cuda.synchronize()
    steps = 200
    t0 = time.perf_counter()
    hist = model.fit(x=samples, y=[ycls, yreg], batch_size=1, verbose=0, epochs=1, steps_per_epoch=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"Model.fit: {steps / dt:.1f} steps/s ({1e3 * dt / steps:.2f} ms/step), loss {hist.history['loss'][-1]:.4f}")
