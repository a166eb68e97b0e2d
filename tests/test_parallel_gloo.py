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


def _probe_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lisec_amd.parallel import DataParallel
    dp = DataParallel("cpu")
    # rank 1 cannot load RCCL (faked): NO rank may go on to the collective lisec_comm_init, all take torch.distributed
    agreed = dp.agree_on_data_plane(rank != 1)
    both_ok = dp.agree_on_data_plane(True)
    grad = torch.full((8,), float(rank + 1))
    dp.average_(grad)                                      # the exchange still works: through torch.distributed
    np.save(os.path.join(out_dir, f"probe{rank}.npy"),
            np.array([float(agreed), float(both_ok), float(dp.comm is None), float(grad[0]), float(dp.rccl_ranks())]))
    dp.close()


def test_rccl_probe_failing_on_one_rank_sends_every_rank_to_torch(tmp_path):
    """ncclCommInitRank is a collective, so a rank without RCCL must be found BEFORE the others enter
    it.  Every rank probes first and the answers are gathered; with rank 1's probe failing every rank must decide
    `torch.distributed` -- nobody is left waiting in lisec_comm_init."""
    port = _free_port()
    mp.spawn(_probe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        agreed, both_ok, no_comm, g, ranks = np.load(tmp_path / f"probe{r}.npy")
        assert agreed == 0.0                               # one bad rank -> False on EVERY rank
        assert both_ok == 1.0                              # and True when all report OK
        assert no_comm == 1.0 and g == 1.5 and ranks == 0.0


def test_device_index_per_local_rank():
    """every rank must resolve to its own GPU (cuda:LOCAL_RANK); sharing one card is only legal with gloo."""
    import pytest
    from lisec_amd.parallel import resolve_device_index
    for r in range(8):
        env = dict(WORLD_SIZE="8", RANK=str(r), LOCAL_RANK=str(r))
        assert resolve_device_index(env, device_count=8) == r
    # single process: torch's current device (-1 = keep), or LISEC_DEVICE
    assert resolve_device_index({}, device_count=1) == -1
    assert resolve_device_index({"LISEC_DEVICE": "3"}, device_count=8) == 3
    # more ranks than GPUs
    with pytest.raises(RuntimeError, match="only 1 GPU"):
        resolve_device_index(dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), device_count=1)
    # every rank pinned to one card: refused with RCCL, allowed with gloo (the one-GPU rehearsal)
    with pytest.raises(RuntimeError, match="one GPU per"):
        resolve_device_index(dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", LISEC_BENCH_DEVICE="0"), device_count=1)
    assert resolve_device_index(dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", LISEC_BENCH_DEVICE="0",
                                     LISEC_DIST_BACKEND="gloo"), device_count=1) == 0


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_self_launch_two_ranks_dry_run():
    """`python bench.py --gpus 2` (no torch.distributed.run): the launcher starts two fresh ranks, they rendezvous
    on 127.0.0.1, rank 0 prints exactly ONE JSON line with the max-over-ranks time (GPU work replaced by a sleep)."""
    r, lines = _run_bench(dict(LISEC_BENCH_DRYRUN="1", LISEC_DIST_BACKEND="gloo"), "--gpus", "2", "--steps", "4",
                          "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 4 and len(j["ms_per_step_per_rank"]) == 2
    # rank 1 sleeps twice as long: the reported step time is the slowest rank's
    assert j["ms_per_step"] == max(j["ms_per_step_per_rank"]) and j["ms_per_step_per_rank"][1] > j["ms_per_step_per_rank"][0]
    assert abs(j["value"] - 2 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies must not leave the launcher (or the other ranks) waiting in the rendezvous."""
    import time
    t0 = time.time()
    r, lines = _run_bench(dict(LISEC_BENCH_DRYRUN="1", LISEC_DIST_BACKEND="gloo", LISEC_BENCH_FAIL_RANK="1"),
                          "--gpus", "2", "--steps", "1")
    assert r.returncode == 3 and not lines
    assert time.time() - t0 < 120                      # rank 0 was taken down, not left to time out in the store
