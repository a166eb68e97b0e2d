"""World-size-2 gloo test of the data-parallel host logic (sharding, gradient averaging, replicas stay
identical).  Runs on CPU; the GPU path only swaps the backend (RCCL) and the scale kernel."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lisec_amd.parallel import DataParallel
    dp = DataParallel("cpu")
    assert (dp.rank, dp.world) == (rank, world)
    samples = list(range(7))
    mine = dp.shard(samples)
    assert mine == samples[rank:6:2]                      # equal counts, tail dropped
    theta = torch.full((1000,), float(rank))
    dp.broadcast_(theta)
    assert (theta == 0).all()
    vel = torch.zeros_like(theta)
    for step, s in enumerate(mine):
        grad = torch.full((1000,), float(s + 1))           # per-sample gradient
        dp.average_(grad)
        lr_t = 0.01 / (1 + 1e-6 * step)
        vel = 0.9 * vel - lr_t * grad
        theta = theta + 0.9 * vel - lr_t * grad
    assert abs(dp.max_float(float(rank)) - (world - 1)) < 1e-12
    dp.barrier()
    np.save(os.path.join(out_dir, f"theta{rank}.npy"), theta.numpy())
    dp.close()


def test_two_rank_gradient_averaging(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "theta0.npy"), np.load(tmp_path / "theta1.npy")
    assert np.array_equal(t0, t1)                          # replicas stay bit-identical
    # reference: sequential SGD-Nesterov on the rank-averaged gradients
    theta, vel = 0.0, 0.0
    for step in range(3):
        g = ((2 * step + 1) + (2 * step + 2)) / 2.0
        lr_t = 0.01 / (1 + 1e-6 * step)
        vel = 0.9 * vel - lr_t * g
        theta = theta + 0.9 * vel - lr_t * g
    assert np.allclose(t0, theta, rtol=1e-6)
