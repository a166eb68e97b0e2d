"""Two data-parallel ranks through the REAL training path (LisecNet on the GPU, gradients averaged with
torch.distributed) -- both ranks share the one GPU of the test box, so the gloo backend carries the
all-reduce (RCCL needs one GPU per rank; the driver's multi-GPU bench uses backend 'nccl')."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SMALL = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)


def _cloud(seed, n=2500):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1)
    return pts.astype(np.float32)


def _targets(seed):
    rng = np.random.default_rng(100 + seed)
    return rng.integers(0, 3, (8, 16, 2)).astype(np.float32), rng.normal(0, 1, (8, 16, 14)).astype(np.float32)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LISEC_DIST_BACKEND="gloo")
    from lisec_amd import model_training as mt
    np.random.seed(0)
    model = mt.createModel(16, 32, 8, 35)                 # WORLD_SIZE=2 -> DataParallel inside, params broadcast
    assert model.dp is not None and model.dp.world == 2
    model.compile(optimizer=mt.optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True), loss=['mse', 'mse'])
    samples = [mt.VFE_preprocessing(_cloud(s), **SMALL) for s in range(4)]
    ys = [_targets(s) for s in range(4)]
    model.fit(x=samples, y=[np.stack([y[0] for y in ys]), np.stack([y[1] for y in ys])], batch_size=1, verbose=0,
              epochs=1, steps_per_epoch=4, shuffle=False)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"theta{rank}.npy"), model.net.params.theta.cpu().numpy())
    model.dp.barrier()
    model.dp.close()


def test_two_ranks_train_identically_and_match_manual_averaging(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "theta0.npy"), np.load(tmp_path / "theta1.npy")
    assert np.array_equal(t0, t1)                            # replicas stay bit-identical

    # single process: same two steps with the per-rank gradients averaged by hand
    from lisec_amd import model_training as mt
    from lisec_amd import ops
    model = mt.createModel(16, 32, 8, 35)
    model.compile(optimizer=mt.optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True), loss=['mse', 'mse'])
    net, dev = model.net, model.net.device
    samples = [mt.VFE_preprocessing(_cloud(s), **SMALL).sample for s in range(4)]
    ys = [_targets(s) for s in range(4)]
    for step in range(2):                                    # steps_per_epoch 4 // world 2
        acc = None
        for rank in range(2):
            i = [0, 2][step] if rank == 0 else [1, 3][step]   # rank r takes samples r, r+2
            net.forward(samples[i], training=True)
            net.backward(torch.from_numpy(ys[i][0]).to(dev), torch.from_numpy(ys[i][1]).to(dev))
            acc = net.grad.clone() if acc is None else acc + net.grad
        net.grad.copy_(acc)
        ops.scale_(net.grad, 0.5)
        net.apply_gradients()
    torch.cuda.synchronize()
    ref = net.params.theta.cpu().numpy()
    assert np.allclose(t0, ref, rtol=1e-5, atol=1e-6)
