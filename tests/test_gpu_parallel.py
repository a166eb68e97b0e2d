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


def _worker(rank, world, port, out_dir, step_plan=True):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LISEC_DIST_BACKEND="gloo", LISEC_BENCH_DEVICE="0",   # both ranks on cuda:0
                      LISEC_TUNING="step_plan=%d" % (1 if step_plan else 0))
    from lisec_amd import model_training as mt
    np.random.seed(0)
    model = mt.createModel(16, 32, 8, 35)                 # WORLD_SIZE=2 -> DataParallel inside, params broadcast
    assert model.dp is not None and model.dp.world == 2
    model.compile(optimizer=mt.optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True), loss=['mse', 'mse'])
    def cloud(s):
        # the plan pads every sweep into its 4096-point buffer with points the voxeliser drops; the Python schedule gets
        # the same padded sweeps (the row-list kernels plan their K slices per capacity: same summation order)
        c = _cloud(s)
        if step_plan:
            return c
        out = np.full((4096, 3), 1.0e6, np.float32)
        out[:len(c)] = c
        return out
    samples = [mt.VFE_preprocessing(cloud(s), **SMALL) for s in range(4)]
    ys = [_targets(s) for s in range(4)]
    model.fit(x=samples, y=[np.stack([y[0] for y in ys]), np.stack([y[1] for y in ys])], batch_size=1, verbose=0,
              epochs=1, steps_per_epoch=4, shuffle=False)
    torch.cuda.synchronize()
    # the data-parallel step replays a step plan too: the gloo exchange rides in it as host calls of the library
    assert (getattr(model, "_captured", None) is not None) == step_plan
    np.save(os.path.join(out_dir, f"theta{rank}.npy"), model.net.params.theta.cpu().numpy())
    np.save(os.path.join(out_dir, f"state{rank}.npy"), model.net.params.state.cpu().numpy())
    # every rank calls save() on the SAME path (as train() does): rank 0 alone writes, the file carries the
    # rank-mean of the per-replica BatchNormalization moving statistics, the replicas keep their own
    model.save(os.path.join(out_dir, "ckpt.npz"))
    assert os.path.exists(os.path.join(out_dir, "ckpt.npz"))       # closed before anyone passes the barrier
    assert np.array_equal(model.net.params.state.cpu().numpy(), np.load(os.path.join(out_dir, f"state{rank}.npy")))
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
    s0, s1 = np.load(tmp_path / "state0.npy"), np.load(tmp_path / "state1.npy")
    assert not np.array_equal(s0, s1)                        # per-replica moving statistics (different samples)
    from lisec_amd import model_training as mt
    ck = mt.load_model(str(tmp_path / "ckpt.npz"))
    assert np.array_equal(ck.net.params.theta.cpu().numpy(), t0)
    assert np.allclose(ck.net.params.state.cpu().numpy(), 0.5 * (s0.astype(np.float64) + s1), rtol=1e-6, atol=1e-7)

    # the same two ranks on the Python schedule (no step plan): bit-identical variables
    eager_dir = tmp_path / "eager"
    eager_dir.mkdir()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port2 = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port2, str(eager_dir), False), nprocs=2, join=True)
    assert np.array_equal(np.load(eager_dir / "theta0.npy"), t0) and np.array_equal(np.load(eager_dir / "state1.npy"), s1)

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


def test_bench_self_launch_two_ranks_on_one_gpu():
    """The exact path the driver's scaling run takes -- plain `python bench.py --gpus 2` -- rehearsed on the one-GPU
    box: the launcher starts two fresh ranks, both pinned to cuda:0 (LISEC_BENCH_DEVICE), gradients averaged over
    gloo instead of RCCL (which needs one GPU per rank).  Same step code, same timing reduction, same JSON."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(LISEC_DIST_BACKEND="gloo", LISEC_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 and j["scaling"] == "weak"
    assert j["dist_backend"] == "gloo" and j["rccl_ranks"] == 0 and len(j["ms_per_step_per_rank"]) == 2
    assert j["value"] > 0 and np.isfinite(j["config"]["final_loss"])
    assert abs(j["value"] - 2 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]


def test_rccl_allreduce_entry_one_rank():
    """lisec_comm_unique_id / lisec_comm_init / lisec_allreduce_grads / lisec_comm_destroy (include/lisec_hip.h 4b)
    straight through ctypes, as a C caller would use them: a one-rank RCCL communicator on this GPU, sum over the
    (single) rank divided by `world`, in place, on an explicit stream."""
    import ctypes
    from lisec_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    ident = ctypes.create_string_buffer(128)
    _lib.check(lib.lisec_comm_unique_id(ident))
    assert any(ident.raw)
    comm = ctypes.c_void_p()
    _lib.check(lib.lisec_comm_init(0, 1, bytes(ident.raw), ctypes.byref(comm)))
    assert comm.value
    rng = np.random.default_rng(0)
    host = rng.normal(0, 1, 6_491_072).astype(np.float32)         # the size of the flat gradient buffer
    x = torch.from_numpy(host).to(dev)
    st = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream())
    count = ctypes.c_int(0)
    _lib.check(lib.lisec_comm_count(comm, ctypes.byref(count)))
    assert count.value == 1
    _lib.check(lib.lisec_comm_probe())
    _lib.check(lib.lisec_allreduce_grads(comm, x.data_ptr(), x.numel(), -4, st.cuda_stream))         # explicit divisor 4
    lo = 1_000_000                                                                                    # a sub-range
    _lib.check(lib.lisec_allreduce_grads(comm, x.data_ptr() + 4 * lo, x.numel() - lo, 1, st.cuda_stream))   # world == count
    _lib.check(lib.lisec_allreduce_grads(comm, x.data_ptr() + 4 * lo, x.numel() - lo, 0, st.cuda_stream))   # 0: the count
    st.synchronize()
    assert np.array_equal(x.cpu().numpy(), host * np.float32(0.25))
    # a caller-supplied world that is not the communicator's size would mis-scale every gradient: refused, nothing enqueued
    assert lib.lisec_allreduce_grads(comm, x.data_ptr(), x.numel(), 4, st.cuda_stream) != 0
    assert b"communicator has 1 ranks" in lib.lisec_last_error()
    st.synchronize()
    assert np.array_equal(x.cpu().numpy(), host * np.float32(0.25))
    assert lib.lisec_allreduce_grads(comm, x.data_ptr(), 6, 1, st.cuda_stream) != 0                   # n % 4 != 0: refused
    assert b"multiple of 4" in lib.lisec_last_error()
    _lib.check(lib.lisec_comm_destroy(comm))


def test_bench_one_rank_through_rccl_data_plane():
    """bench.py with LISEC_FORCE_DP=1: one rank, backend nccl, and the gradient exchange of every step going through
    lisec_allreduce_grads on the C ABI's own RCCL communicator (both buckets) -- the code the N-GPU run executes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LISEC_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("LISEC_DIST_BACKEND", None)
    def run(extra_env):
        e = dict(env)
        e.update(extra_env)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3",
                            "--no-cpu-baseline"], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    j = run({})
    assert j["dist_backend"] == "nccl" and j["rccl_ranks"] == 1
    assert j["gradient_exchange"] == "lisec_allreduce_grads (RCCL)"
    assert np.isfinite(j["config"]["final_loss"]) and j["value"] > 0
    # the exchange is part of the recorded step: same C-side schedule as the one-GPU line, both buckets included ...
    assert j["config"]["launch"].startswith("step plan")
    # ... bit-identical to the Python schedule of the same data-parallel step (the plain recorded form: the pipelined one
    # trains one step more before the timed region, so its final loss is another step's) ...
    r_ = run({"LISEC_TUNING": "pipeline_voxels=0"})
    e = run({"LISEC_TUNING": "step_plan=0"})
    assert r_["config"]["launch"].startswith("step plan") and e["config"]["launch"].startswith("Python schedule")
    assert e["config"]["final_loss"] == r_["config"]["final_loss"]
    # ... and as fast as the plain one-GPU line (a one-rank all-reduce of 26 MB in two buckets on its own stream)
    env2 = {k: v for k, v in env.items() if k != "LISEC_FORCE_DP"}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       env=env2, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    plain = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        with open(os.path.join(root, "gpurun_out", "dp_plan_vs_plain.txt"), "a") as f:
            f.write(f"one rank through RCCL, step plan: {j['ms_per_step']:.4f} ms; Python schedule: {e['ms_per_step']:.4f} ms; "
                    f"plain one-GPU line: {plain['ms_per_step']:.4f} ms\n")
    assert j["ms_per_step"] <= 1.05 * plain["ms_per_step"], (j["ms_per_step"], plain["ms_per_step"])
