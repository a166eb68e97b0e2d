#!/usr/bin/env python3
"""Headline benchmark: Lyft-grid samples/s, forward + backward + SGD step (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 5
    python bench.py --gpus N --steps K --warmup W         (starts the N ranks itself, one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one synthetic lidar sweep per GPU, inputs resident
in HBM: voxelise (U20k cloud) -> sparse-exact VFE -> 3 Conv3D middle layers -> RPN ->
MSE+MSE loss (the reference's compile(), model_training.py:296) -> full backward -> (N > 1: RCCL
all-reduce of the 6.49 M fp32 gradients) -> SGD-Nesterov update.  Weak scaling: one sample per GPU.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel = the second middle Conv3D in its Winograd form (k_wino, fp32 MFMA bound; forward launch), timed
                live with HIP events on the launch stream; clock and HBM traffic of that kernel come from the PMC passes
                summarised in profiles/r03_pmc_dominant.json (tools/pmc_summary.py writes it; null when absent)
  roofline_vfe  the VFE grid writer (HBM bound)
  r200k, smoothl1_ce   the same step on a 200 000-point sweep / with BASELINE config 4's loss pair
  cpu_baseline  the dense torch-CPU oracle (port of the reference's dense Keras graph) timed on this
                host, on a stated shrunken grid, extrapolated by dense row count.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.3 TB/s achievable per the same guide)


def pmc_dominant():
    """Counters of the dominant kernel's own launch from the committed PMC summary (rocprofv3 --pmc passes of this file,
    one counter set per pass, gfx950 correction FETCH_SIZE x2 applied by tools/pmc_summary.py): clock_ghz, traffic_bytes."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_dominant.json")))
    for path in reversed(paths):                     # the newest committed round
        try:
            with open(path) as f:
                d = json.load(f)
            d.setdefault("file", os.path.relpath(path, ROOT))
            return d
        except (OSError, ValueError):
            continue
    return {}


def u20k_cloud(seed, n=20000):
    rng = np.random.default_rng(seed)
    p = np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1)
    return p.astype(np.float32)


def r200k_cloud(seed, n=200000):
    """Lyft-like sweep (SURVEY 8d 'Cloud R200k'): dense near the sensor, voxels far beyond 35 points."""
    rng = np.random.default_rng(seed)
    az = rng.uniform(0, 2 * np.pi, n)
    r = 2.0 + 68.0 * rng.uniform(0, 1, n) ** 2
    return np.stack([r * np.cos(az), r * np.sin(az), rng.uniform(-0.2, 2.2, n)], 1).astype(np.float32)


def synthetic_targets(seed, Ho, Wo):
    """cls in {0,1,2} with 256 valid anchors, reg ~ N(0,1) on positives."""
    rng = np.random.default_rng(1000 + seed)
    cls = np.zeros((Ho * Wo * 2,), np.float32)
    idx = rng.choice(Ho * Wo * 2, 256, replace=False)
    cls[idx] = rng.integers(1, 3, 256)
    reg = np.zeros((Ho * Wo, 14), np.float32)
    pos = np.unique(idx // 2)
    reg[pos] = rng.normal(0, 1, (len(pos), 14))
    return cls.reshape(Ho, Wo, 2), reg.reshape(Ho, Wo, 14)


def event_time_ms(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def voxelizer_leg(vox, dev, skip_cpu=False):
    """BASELINE.md plan item 1: the HIP voxeliser (lisec_voxelize: 7 launches -- zero, key + count, three scan kernels,
    place, features) on both synthetic sweeps,
    points/s and algorithmic bytes/s against the HBM peak, next to the CPU restatement (oracle/voxel_ref.py, numpy,
    one core) on the same U20k cloud and the reference's own Python loop (2.62 s for 20 000 points measured in the
    build container, "~15 sec" for ~200 000 in its source, model_training.py:116)."""
    out = {}
    for name, cloud in (("u20k", u20k_cloud(0)), ("r200k", r200k_cloud(0))):
        pts = torch.from_numpy(cloud).to(dev)
        ms = event_time_ms(lambda: vox(pts), 30)
        hi = vox(pts).host_info()
        ncells = vox.ncells
        # algorithmic bytes: 12 n read + 24 rows written + the 4-byte cell counters zeroed, counted and
        # scanned (3 passes) + cell_voxel written + 16 B of per-voxel metadata
        byt = 12.0 * len(cloud) + 24.0 * hi["rows"] + 4.0 * ncells * 4 + 16.0 * hi["V"]
        out[name] = dict(points=int(len(cloud)), voxels=hi["V"], rows=hi["rows"], us_per_sweep=ms * 1e3,
                         points_per_s=len(cloud) / (ms * 1e-3), achieved_gbs=byt / (ms * 1e-3) / 1e9,
                         frac_hbm=byt / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, bytes_per_sweep=byt)
    if not skip_cpu:
        from oracle import voxel_ref
        from lisec_amd import Constants
        cfg = dict(xSize=Constants.voxelx, ySize=Constants.voxely, zSize=Constants.voxelz, sampleSize=Constants.maxPoints,
                   maxVoxelX=Constants.nx // 2, maxVoxelY=Constants.ny // 2, maxVoxelZ=Constants.nz)
        cloud = u20k_cloud(0).astype(np.float64)
        t0 = time.perf_counter()
        reps = 0
        while reps < 3 or time.perf_counter() - t0 < 2.0:
            voxel_ref.voxelize_ref(cloud, **cfg)
            reps += 1
        sec = (time.perf_counter() - t0) / reps
        out["cpu_restatement"] = dict(points_per_s=len(cloud) / sec, s_per_sweep=sec, cores=1, kind="port",
                                      sample=f"oracle/voxel_ref.voxelize_ref (numpy), U20k, {reps} sweeps")
    secs = []
    try:
        for ln in open(os.path.join(ROOT, "tests", "golden", "voxel_goldens_timing.txt")):
            f = ln.split()
            if len(f) == 5 and f[0].startswith("voxel_u20k"):
                secs.append(float(f[4]))
    except OSError:
        pass
    ref_s = sum(secs) / len(secs) if secs else 2.62
    out["reference"] = dict(s_per_sweep_20k=ref_s, points_per_s=20000 / ref_s,
                            note="serialize_data.VFE_preprocessing (pure-Python loop) run unmodified on the U20k "
                                 "fixtures in the build container, one core (tests/golden/voxel_goldens_timing.txt; "
                                 "2.62 s in the survey's first probe); '~15 sec' at ~200 000 points per its own "
                                 "source (model_training.py:116)")
    return out


def host_ram_gb():
    """(total, available) host RAM in GB from /proc/meminfo (printed beside the core count)."""
    tot = avail = None
    try:
        for ln in open("/proc/meminfo"):
            f = ln.split()
            if f[0] == "MemTotal:":
                tot = int(f[1]) / 1e6
            elif f[0] == "MemAvailable:":
                avail = int(f[1]) / 1e6
    except OSError:
        pass
    return tot, avail


def cpu_baseline(seconds_budget=12.0):
    """Dense torch-CPU oracle (fwd + bwd + update): at the FULL Lyft grid when the host has the RAM for the dense rank-6
    tensors (BASELINE.md plan item 3: >= ~96 GB; one step), otherwise on a shrunken grid, extrapolated by dense row count."""
    from oracle import model_ref as M
    from oracle import voxel_ref
    ram_total, ram_avail = host_ram_gb()
    if ram_avail is not None and ram_avail >= 160.0 and os.environ.get("LISEC_CPU_BASELINE_GRID", "full") == "full":
        try:
            return cpu_baseline_grid(M, voxel_ref, 8, 200, 400, 20000, 1, ram_total, ram_avail)
        except MemoryError:
            pass
    return cpu_baseline_grid(M, voxel_ref, 8, 48, 96, None, None, ram_total, ram_avail, seconds_budget)


def cpu_baseline_grid(M, voxel_ref, D, H, W, n_points, max_steps, ram_total, ram_avail, seconds_budget=12.0):
    # the GPU box gives one job a 16-core share of a much larger host: more threads only oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    # shrunken: 8 x 48 x 96 = 1/17.4 of the Lyft grid's dense rows: ~2 s per step on 16 cores, ~5 GB of RAM
    full = (H, W) == (200, 400)
    cfg = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=H // 2, maxVoxelY=W // 2, maxVoxelZ=8)
    rng = np.random.default_rng(0)
    n = n_points or 20000 * (H * W) // (200 * 400)
    ext = 0.5 * (H // 2) * 1.1          # the same 10 % of points beyond the grid as the U20k cloud
    pts = np.stack([rng.uniform(-ext, ext, n), rng.uniform(-ext, ext, n), rng.uniform(-0.5, 2.5, n)], 1)
    vox = voxel_ref.voxelize_ref(pts.astype(np.float32).astype(np.float64), **cfg)
    dense = torch.from_numpy(voxel_ref.to_dense(vox, (D, H, W, 35, 6)))[None]
    p = M.glorot_params()
    vel = {nme: torch.zeros_like(p[nme]) for nme, _, k in M.param_specs() if M.is_trainable(k)}
    yc = torch.zeros(1, H // 2, W // 2, 2)
    yr = torch.zeros(1, H // 2, W // 2, 14)
    t0 = time.time()
    steps = 0
    while True:
        _, _, p, vel, _ = M.train_step(p, vel, dense, yc, yr, steps)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
        if steps >= 2 and (time.time() - t0 > seconds_budget or steps >= 6):
            break
    per_step = (time.time() - t0) / steps
    scale = (200 * 400) / float(H * W)
    ram = dict(ram_total_gb=round(ram_total, 1) if ram_total else None, ram_available_gb=round(ram_avail, 1) if ram_avail else None)
    if full:
        return dict(value=1.0 / per_step, unit="samples/s", cores=cores, kind="port", extrapolated=False,
                    sample=f"dense torch-CPU oracle, fwd+bwd+SGD, the FULL grid {D}x{H}x{W}x35x6 ({steps} step, "
                           f"{per_step:.1f} s/step, {cores} threads)", **ram)
    return dict(value=1.0 / (per_step * scale), unit="samples/s", cores=cores, kind="port", extrapolated=True,
                sample=f"dense torch-CPU oracle, fwd+bwd+SGD, grid {D}x{H}x{W}x35x6 ({steps} steps, "
                       f"{per_step:.2f} s/step), extrapolated x{scale:.1f} by dense row count to 8x200x400 "
                       f"(full grid needs >= 160 GB of available host RAM)", **ram)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh child interpreters of this same command, one rank
    per GPU (RANK = LOCAL_RANK = 0..N-1, rendezvous on 127.0.0.1), BEFORE anything in this process touches the GPU.
    Rank 0 inherits stdout (it prints the one JSON line); the other ranks' stdout goes to stderr.  Returns the exit code:
    0 when every rank returned 0.  A rank that fails takes the others down (they would wait in the rendezvous forever)."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:                               # exact PIDs we started, never a pattern
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def dry_run(rank, world, args):
    """LISEC_BENCH_DRYRUN=1: the launcher / rendezvous / timing-reduction / JSON path of this file with the GPU work
    replaced by a sleep (CPU test of `python bench.py --gpus N`; gloo)."""
    from lisec_amd.parallel import DataParallel
    if os.environ.get("LISEC_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
        raise SystemExit(3)
    dp = DataParallel("cpu") if world > 1 else None
    if dp is not None:
        dp.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps * (1 + rank))
    dt = time.perf_counter() - t0
    per_rank = dp.gather_floats(dt) if dp is not None else [dt]
    if dp is not None:
        dt = dp.max_float(dt)
    if rank == 0:
        print(json.dumps({"metric": "lyft_samples_per_sec_fwd_bwd", "value": world * args.steps / dt, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "ms_per_step_per_rank": [1e3 * t / args.steps for t in per_rank], "dry_run": True,
                          "dist_backend": dp.backend_name() if dp is not None else None,
                          "rccl_ranks": dp.rccl_ranks() if dp is not None else 0}), flush=True)
    if dp is not None:
        dp.barrier()
        dp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # 100 steps = 0.56 s of GPU time at 5.6 ms per step
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss", choices=["mse", "smoothl1_ce"], default="mse",
                    help="mse = the reference's compile(loss=['mse','mse']) (model_training.py:296); smoothl1_ce = the "
                         "sigmoid cross-entropy + SmoothL1 pair BASELINE.json's config 3 names (same kernels otherwise)")
    ap.add_argument("--cloud", choices=["u20k", "r200k"], default="u20k",
                    help="u20k: the headline workload (BASELINE configs 2-4); r200k: a Lyft-size sweep")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if os.environ.get("LISEC_BENCH_DRYRUN") == "1":
        return dry_run(rank, world, args)
    # LISEC_BENCH_DEVICE pins every rank to one device index (rehearsing the N > 1 code path on a one-GPU box
    # together with LISEC_DIST_BACKEND=gloo); the driver's multi-GPU run uses cuda:LOCAL_RANK and RCCL
    from lisec_amd.parallel import select_device
    dev = select_device(local_rank=local_rank, world=world)

    from lisec_amd import Constants, ops
    from lisec_amd.network import LisecNet
    from lisec_amd.parallel import DataParallel
    from lisec_amd.voxelizer import Voxelizer

    # LISEC_FORCE_DP=1 exercises the RCCL path (init, broadcast, all-reduce, barrier) even with one rank
    dp = DataParallel(dev) if (world > 1 or os.environ.get("LISEC_FORCE_DP") == "1") else None
    net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints, device=dev)
    vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints,
                    Constants.nx // 2, Constants.ny // 2, Constants.nz, device=dev)
    cloud = u20k_cloud(rank) if args.cloud == "u20k" else r200k_cloud(rank)
    pts = torch.from_numpy(cloud).to(dev)
    ycls, yreg = synthetic_targets(rank, net.Ho, net.Wo)
    ycls, yreg = torch.from_numpy(ycls).to(dev), torch.from_numpy(yreg).to(dev)
    if dp is not None:
        dp.broadcast_(net.params.theta)
        dp.broadcast_(net.params.state)
        net.params.touch()
    allreduce = dp.bucketed() if dp is not None else None
    # One GPU: the whole step (voxelise + forward + backward + SGD + the repack for the next step) is recorded once as a
    # step plan of the C ABI and re-issued by ONE call per step (lisec_step_plan_run: the eager launches on the same two
    # streams, without the Python schedule in front of each of them) -- with N > 1 ranks too: the two-bucket gradient
    # exchange is part of the recorded schedule.  LISEC_TUNING=step_plan=0 issues each step from the Python schedule.
    from lisec_amd import _lib
    use_plan = _lib.knob("step_plan", True)       # data parallel too: the gradient exchange is part of the recorded schedule

    if use_plan and _lib.knob("pipeline_voxels", True):
        # ... with the NEXT sweep's voxelisation inside the step (second stream, under the backward pass): every step still
        # voxelises one sweep and trains on one
        from lisec_amd.network import PipelinedStep
        captured = PipelinedStep(net, vox, len(cloud), dtype=pts.dtype, loss=args.loss, allreduce=allreduce)
        captured.prime(pts, ycls, yreg)         # inputs resident in HBM before the timed region, as in the eager path
        captured.stage_next(pts, ycls, yreg)
        captured.step()
        captured.stage_next(pts, ycls, yreg)    # both buffer sets hold the sweep
        step = captured.step
    elif use_plan:
        from lisec_amd.network import RecordedStep
        captured = RecordedStep(net, vox, len(cloud), dtype=pts.dtype, loss=args.loss, allreduce=allreduce)
        captured.load(pts, ycls, yreg)          # inputs resident in HBM before the timed region, as in the eager path
        step = captured.replay
    else:
        def step():
            sample = vox(pts)
            return net.train_step(sample, ycls, yreg, loss=args.loss, allreduce=allreduce)

    torch.cuda.synchronize()
    if True:
        for _ in range(args.warmup):
            step()
        if dp is not None:
            dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = dp.gather_floats(dt) if dp is not None else [dt]
    if dp is not None:
        dt = dp.max_float(dt)
    loss_val = float(loss[0].item())

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel, timed live on the launch stream ----------------------
        # Since the first Conv3D runs over the VFE's compact output (csrc/field_conv.hip, ~1 GFLOP), the largest contraction
        # of the step is the second middle block's (2 * 160 000 positions * 27 taps * 64 * 64 = 35.4 GFLOP forward, the same
        # again for each gradient).  Its forward and data gradient run in the Winograd F(2x2, 3x3) form (csrc/wino.hip, 4/9 of
        # the multiplications); k_wino is the symbol with the most time in the step and the forward its longest launch.
        mid2 = next(L for L in net.layers if L["name"] == "mid2")
        c2, dg2 = mid2["conv"], net.dgeom[mid2["conv"].name]
        dz2, du1 = net.dact["mid2.z"], net.dact["mid1.u"]
        flops = 2.0 * c2.M * 27 * 64 * 64
        pmc = pmc_dominant()
        pmc_src = "PMC passes of this file: " + pmc.get("source", "profiles/r*_pmc_dominant.json absent")

        def run_mid2_dgrad_direct():
            # TAG_ROOFLINE: one un-sliced launch under its own symbol (k_igemm_halo<1,false,1,2,64,false>), so the row of that
            # symbol in the rocprofv3 --stats summary of this command is this layer alone
            ops.conv_forward(dg2, dz2, net.packed_t[c2.name][0], du1, flags=ops.TAG_ROOFLINE)
        ms_direct = event_time_ms(run_mid2_dgrad_direct, 20)
        if c2.name in net.packed_wu:
            def run_mid2_forward_winograd():
                # TAG_ROOFLINE: the same kernel under the symbol k_wino<false,0,1>
                ops.conv_forward_winograd(c2.g, net.act["mid1.u"], net.packed_wu[c2.name], net.act["mid2.y"],
                                          bias=net.params.view(c2.bias), flags=ops.TAG_ROOFLINE)
            ms = event_time_ms(run_mid2_forward_winograd, 20)
            tf = flops / (ms * 1e-3) / 1e12
            # algorithmic FLOPs = the contraction's own (every (position, tap, channel pair) once); the Winograd
            # form EXECUTES 16 / 36 of them as MFMA work on 650 blocks of 64 tiles (40 000 tiles padded to 41 600)
            share = (16.0 / 36.0) * (650.0 * 64.0 / (c2.M / 4.0))
            roofline = dict(bound="mfma", kernel="k_wino<false,0,1> mid2 Conv3D 64->64 k3 s1 forward, Winograd F(2x2,3x3) form",
                            achieved=tf, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_F32_MFMA_TFLOPS,
                            frac_executed=tf * share / PEAK_F32_MFMA_TFLOPS, executed_share=share,
                            note="achieved = algorithmic FLOPs of the contraction / time; the kernel executes executed_share of "
                                 "them on the matrix cores (frac_executed = that work against the same peak)",
                            clock_ghz=pmc.get("clock_ghz"), clock_note="GRBM_GUI_ACTIVE/8/duration, " + pmc_src,
                            traffic=pmc.get("traffic_bytes"), traffic_unit="bytes/launch, " + pmc_src,
                            mfma_busy=pmc.get("mfma_busy"),
                            us_per_launch=ms * 1e3, flops_per_launch=flops,
                            in_step_us=pmc.get("in_step_us"),
                            in_step_frac=(flops / (pmc["in_step_us"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS
                                          if pmc.get("in_step_us") else None))
        else:                                        # LISEC_TUNING=winograd=0: the direct data gradient is the dominant launch
            tf = flops / (ms_direct * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel="k_igemm_halo<1,false,1,2,64,false> mid2 Conv3D 64->64 k3 s1 data gradient", achieved=tf,
                            peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_F32_MFMA_TFLOPS,
                            frac_executed=tf / PEAK_F32_MFMA_TFLOPS, executed_share=1.0,
                            clock_ghz=None, traffic=None, us_per_launch=ms_direct * 1e3, flops_per_launch=flops)
        # the other large contractions, same clock: mid2 forward, and the dense form of the first Conv3D (the dominant
        # kernel of rounds 1-2; sweeps beyond LISEC_FIELD_MAX_VOXELS still take it)
        grid = net.dense_grid()
        mid1 = next(L for L in net.layers if L["name"] == "mid1")
        c1 = mid1["conv"]
        ms_f2 = event_time_ms(lambda: ops.conv_forward(c2.g, net.act["mid1.u"], net.packed[c2.name], net.act["mid2.y"],
                                                       bias=net.params.view(c2.bias), stats=net.parts), 20)
        ms_f1 = event_time_ms(lambda: ops.conv_forward(c1.g, grid, net.packed[c1.name], net.act["mid1.y"],
                                                       bias=net.params.view(c1.bias), stats=net.parts), 10)
        vout, delta = net.vfe.saved_field("vout"), net.vfe.saved_field("delta")
        ms_field = event_time_ms(lambda: ops.conv_field_forward(c1.g, vout, delta, net.vfe._sample, net.packed[c1.name],
                                                                net.act["mid1.y"], net.field_ws,
                                                                bias=net.params.view(c1.bias)), 20)
        f1 = 2.0 * c1.M * 27 * 64 * 64
        roofline["others"] = {
            "mid2_data_gradient_direct": dict(kernel="k_igemm_halo<1,false,1,2,64,false>", us_per_launch=ms_direct * 1e3,
                                              achieved=flops / (ms_direct * 1e-3) / 1e12,
                                              frac=flops / (ms_direct * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, flops_per_launch=flops,
                                              note="the direct (implicit-GEMM) form of the same contraction's data gradient: the "
                                                   "dominant launch of rounds 3-4a, executed == algorithmic"),
            "mid2_forward_direct": dict(us_per_launch=ms_f2 * 1e3, achieved=flops / (ms_f2 * 1e-3) / 1e12,
                                        frac=flops / (ms_f2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, flops_per_launch=flops),
            "mid1_dense_form": dict(us_per_launch=ms_f1 * 1e3, achieved=f1 / (ms_f1 * 1e-3) / 1e12,
                                    frac=f1 / (ms_f1 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, flops_per_launch=f1),
            "mid1_field_form": dict(us_per_call=ms_field * 1e3, launches=2,
                                    note="what the step runs: k_field_taps + k_field_combine over the VFE's compact output, "
                                         "same values as mid1_dense_form"),
        }
        sample = vox(pts)

        def run_vfe():
            net.vfe.forward(sample, True, out=grid)
        ms_v = event_time_ms(run_vfe, 20)
        hi = sample.host_info()
        vfe_bytes = 12.0 * len(cloud) + 24.0 * hi["rows"] + 4.0 * 64 * net.D * net.H * net.W
        ms_vc = event_time_ms(lambda: net.vfe.forward(sample, True, dense=False), 20)
        vfe_compact_bytes = 12.0 * len(cloud) + 24.0 * hi["rows"] + 2.0 * 256 * (hi["V"] + 1)
        ms_g = event_time_ms(lambda: net.vfe.rewrite_grid(grid), 50)
        grid_bytes = 4.0 * 64 * net.D * net.H * net.W
        gbs = grid_bytes / (ms_g * 1e-3) / 1e9
        # traffic from the committed PMC passes (WRITE_SIZE + 2 x FETCH_SIZE of k_vfe_grid's launches WITH the dense grid)
        vfe_pmc = pmc.get("vfe_grid", {})
        roofline_vfe = dict(bound="hbm", kernel="k_vfe_grid (dense (8,200,400,64) VFE output writer; lisec_vfe_forward with a grid)", achieved=gbs,
                            peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS,
                            traffic=vfe_pmc.get("traffic_bytes"),
                            traffic_unit="bytes/launch, " + pmc.get("file", "profiles/r*_pmc_dominant.json absent") + " (vfe_grid)",
                            us_per_launch=ms_g * 1e3, bytes_per_launch=grid_bytes,
                            whole_vfe_forward=dict(us_per_call=ms_v * 1e3, bytes_per_call=vfe_bytes,
                                                   achieved=vfe_bytes / (ms_v * 1e-3) / 1e9,
                                                   frac=vfe_bytes / (ms_v * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                   launches=3),
                            in_step=dict(us_per_call=ms_vc * 1e3, bytes_per_call=vfe_compact_bytes, launches=3,
                                         note="what the training step runs: the same three launches without the dense "
                                              "grid (per-voxel outputs only, read by the field form of the first Conv3D); "
                                              "latency-bound, not an HBM roofline case"))
        voxelizer = voxelizer_leg(vox, dev, args.no_cpu_baseline)
        def side_leg(points_np, loss_name, steps=10):
            """The same step on another sweep / loss pair, `steps` timed steps after 2 warm-up steps."""
            p2 = torch.from_numpy(points_np).to(dev)
            if use_plan and _lib.knob("pipeline_voxels", True):
                from lisec_amd.network import PipelinedStep
                rec = PipelinedStep(net, vox, len(points_np), dtype=p2.dtype, loss=loss_name)
                rec.prime(p2, ycls, yreg)
                rec.stage_next(p2, ycls, yreg)
                rec.step()
                rec.stage_next(p2, ycls, yreg)
                run = rec.step
            elif use_plan:
                from lisec_amd.network import RecordedStep
                rec = RecordedStep(net, vox, len(points_np), dtype=p2.dtype, loss=loss_name)
                rec.load(p2, ycls, yreg)
                run = rec.replay
            else:
                rec, run = None, (lambda: net.train_step(vox(p2), ycls, yreg, loss=loss_name))
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t1) / steps
            if rec is not None:
                rec.close()
            return dict(value=1.0 / d2, unit="samples/s", ms_per_step=1e3 * d2, points=int(p2.shape[0]),
                        voxels=vox(p2).host_info()["V"], steps=steps, loss=loss_name)
        r200k = other_loss = None
        if args.cloud == "u20k" and world == 1:
            # real Lyft sweeps are ~200 000 points (model_training.py:116): the same step on the R200k sweep
            r200k = side_leg(r200k_cloud(rank), args.loss)
            # the loss pair BASELINE config 4 names, next to the reference's own ['mse','mse'] (model_training.py:296)
            other = "smoothl1_ce" if args.loss == "mse" else "mse"
            other_loss = side_leg(cloud, other)
        result = {
            "metric": "lyft_samples_per_sec_fwd_bwd", "value": world * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_per_rank": [1e3 * t / args.steps for t in per_rank],
            "dist_backend": dp.backend_name() if dp is not None else None,
            "rccl_ranks": dp.rccl_ranks() if dp is not None else 0,
            "gradient_exchange": dp.exchange_name() if dp is not None else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Lyft grid 8x200x400x35, {args.cloud.upper()} synthetic cloud, 1 sample/GPU/step, "
                                   "voxelise+VFE+3xConv3D+RPN fwd+bwd, "
                                   + ("MSE+MSE" if args.loss == "mse" else "sigmoid-CE+SmoothL1") + ", SGD-Nesterov",
                       "global_batch": world, "parallelism": f"dp{world}", "points_per_sample": int(len(cloud)),
                       "launch": ("step plan (lisec_step_plan_run: %d recorded launches / event edges per step)" % captured.launches)
                                 if use_plan else "Python schedule (one ctypes call per launch)",
                       "voxels": hi["V"], "final_loss": loss_val},
            "roofline": roofline, "roofline_vfe": roofline_vfe, "voxelizer": voxelizer, "r200k": r200k,
        }
        if other_loss is not None:
            result[other_loss["loss"]] = other_loss
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if dp is not None:
        dp.barrier()
        dp.close()


if __name__ == "__main__":
    main()
