"""Data parallelism over whole samples: one process per GPU, RCCL all-reduce of the flat gradient.

The reference has no distributed code (single process, CPU).  Its fit(batch_size=1) semantics are kept
per replica (BatchNormalization statistics are per-sample, never synchronised); the only exchange per
step is ONE all-reduce of the contiguous 6.49 M-float gradient buffer (26 MB) over xGMI, followed by
identical SGD-Nesterov updates on every rank.

Two planes.  Control (rendezvous, parameter broadcast, barriers, timing reductions): torch.distributed,
backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests of this host logic.  Data (the gradient
exchange of every step): the C ABI's lisec_allreduce_grads on an RCCL communicator of its own
(lisec_comm_unique_id / lisec_comm_init; the 128-byte id travels over the control plane), enqueued on a
dedicated HIP stream -- exactly what a C caller of include/lisec_hip.h would do.  LISEC_ALLREDUCE=torch
routes the exchange through torch.distributed.all_reduce instead (gloo runs always do).
"""
import ctypes
import os

import torch
import torch.distributed as dist


def resolve_device_index(env, device_count=None):
    """Which GPU a rank uses: cuda:LOCAL_RANK, one process per GPU.  LISEC_BENCH_DEVICE overrides it for every rank
    (several ranks on ONE card: only legal with the gloo backend, which tests use on a one-GPU box).  Pure function
    of `env` (a mapping) so that it can be tested without a GPU."""
    world = int(env.get("WORLD_SIZE", "1"))
    local_rank = int(env.get("LOCAL_RANK", env.get("RANK", "0") if world > 1 else "0"))
    override = env.get("LISEC_BENCH_DEVICE")
    backend = env.get("LISEC_DIST_BACKEND") or "nccl"
    if override is not None and override != "":
        if world > 1 and backend == "nccl":
            raise RuntimeError("LISEC_BENCH_DEVICE puts every rank on one GPU; RCCL (backend nccl) needs one GPU per "
                               "rank -- set LISEC_DIST_BACKEND=gloo for a shared-card rehearsal")
        idx = int(override)
    else:
        idx = local_rank if world > 1 else int(env.get("LISEC_DEVICE", "-1"))
    if device_count is not None and idx >= device_count:
        raise RuntimeError(f"rank with LOCAL_RANK={local_rank} wants cuda:{idx} but only {device_count} GPU(s) are "
                           "visible: launch at most one rank per GPU")
    return idx


def select_device(local_rank=None, world=None):
    """Binds this process to its GPU (torch.cuda.set_device) BEFORE any buffer is allocated and returns the
    torch.device.  Single-process runs keep torch's current device unless LISEC_DEVICE names one."""
    env = dict(os.environ)
    if local_rank is not None:
        env["LOCAL_RANK"] = str(local_rank)
    if world is not None:
        env["WORLD_SIZE"] = str(world)
    idx = resolve_device_index(env, torch.cuda.device_count())
    if idx < 0:
        idx = torch.cuda.current_device()
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


class DataParallel:
    def __init__(self, device, backend=None):
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self._own_group = False
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                # LISEC_DIST_BACKEND=gloo lets several ranks share ONE GPU in tests (RCCL wants a GPU per rank)
                backend = os.environ.get("LISEC_DIST_BACKEND") or ("nccl" if self.on_gpu else "gloo")
            self.backend = backend
            kwargs = {}
            if self.on_gpu and backend == "nccl":
                kwargs["device_id"] = self.device
            dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")),
                                    world_size=int(os.environ.get("WORLD_SIZE", "1")), **kwargs)
            self._own_group = True
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if self.on_gpu and self.world > 1 and dist.get_backend() == "nccl":
            # one GPU per rank: two ranks of one host on the same card make RCCL fail late and obscurely
            mine = (os.uname().nodename, self.device.index)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine)
            if len(set(everyone)) != self.world:
                raise RuntimeError(f"data-parallel ranks share a GPU: {everyone}; each rank must own cuda:LOCAL_RANK")
        self.comm = None                 # lisec_comm_t (RCCL communicator) of the data plane
        self.comm_stream = None
        self._comm_ranks = 0
        if self.on_gpu and dist.get_backend() == "nccl" and os.environ.get("LISEC_ALLREDUCE", "rccl") != "torch":
            # lisec_comm_init is a COLLECTIVE (ncclCommInitRank): a rank that cannot load RCCL must be found BEFORE the
            # others enter it, or they would wait in it for ever.  So every rank probes first (no collective, no device
            # work), the answers are gathered over the control plane, and the communicator is only made when all of
            # them can: otherwise EVERY rank takes torch.distributed for the exchange.  A failure after that point is
            # an error (no silent fallback: the ranks could no longer agree on it).
            if self.agree_on_data_plane(self._probe_rccl()):
                self._init_comm()

    @staticmethod
    def _probe_rccl():
        try:
            from . import _lib
            return _lib.load().lisec_comm_probe() == 0
        except Exception:                          # noqa: BLE001 -- a missing library is an answer, not a crash
            return False

    def agree_on_data_plane(self, mine_ok):
        """True when EVERY rank reported that it can make the RCCL communicator (one all_gather over the control plane);
        prints once per rank why the exchange goes through torch.distributed otherwise."""
        flags = [bool(mine_ok)]
        if self.world > 1:
            flags = [None] * self.world
            dist.all_gather_object(flags, bool(mine_ok))
        if all(flags):
            return True
        import sys
        bad = [r for r, ok in enumerate(flags) if not ok]
        print(f"[lisec_amd] rank {self.rank}: RCCL cannot be loaded on rank(s) {bad}; every rank exchanges gradients through "
              "torch.distributed.all_reduce", file=sys.stderr, flush=True)
        return False

    def _init_comm(self):
        from . import _lib
        lib = _lib.load()
        ident = ctypes.create_string_buffer(128)                 # LISEC_COMM_ID_BYTES
        box = [None]
        if self.rank == 0:
            if lib.lisec_comm_unique_id(ident) == 0:             # on failure every rank learns it (None) and falls back
                box = [bytes(ident.raw)]
            else:
                box = [None, lib.lisec_last_error().decode()]
        if self.world > 1:
            box = box + [None] * (2 - len(box))
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise _lib.LisecError("lisec_comm_unique_id failed on rank 0: " + str(box[1] if len(box) > 1 else ""))
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.lisec_comm_init(self.rank, self.world, box[0], ctypes.byref(comm)))
        self.comm = comm
        count = ctypes.c_int(0)
        _lib.check(lib.lisec_comm_count(comm, ctypes.byref(count)))
        if count.value != self.world:
            raise _lib.LisecError(f"the RCCL communicator has {count.value} ranks, torch.distributed {self.world}")
        self._comm_ranks = count.value
        # The exchange stream.  ROCm multiplexes same-priority streams onto a few hardware queues: at the main stream's
        # priority this stream shared ITS queue, and the exchange's wait for the second stream's event held the main
        # stream's launches back behind it (one rank through RCCL: 5.2 ms per step against 4.2).  A priority level of its
        # own -- the lowest: the collective has the whole rest of the backward pass to finish -- gets its own queue.
        least, _greatest = _lib.stream_priority_range()
        prio = _lib.knob("comm_priority", least)
        with torch.cuda.device(self.device):
            self.comm_stream = (torch.cuda.ExternalStream(_lib.create_stream(prio), device=self.device) if prio > 0
                                else torch.cuda.Stream(device=self.device, priority=prio))
        # fork / join edges of the exchange go through the library (lisec_event_record), so that a step plan records them
        self._ev_tail, self._ev_head, self._ev_done = _lib.DeviceEvent(), _lib.DeviceEvent(), _lib.DeviceEvent()

    def exchange_name(self):
        """What carries the gradient exchange: 'lisec_allreduce_grads (RCCL)' or 'torch.distributed.<backend>'."""
        return "lisec_allreduce_grads (RCCL)" if self.comm is not None else "torch.distributed." + dist.get_backend()

    def backend_name(self):
        return dist.get_backend()

    def rccl_ranks(self):
        """Ranks of the communicator that MOVES the gradients: ncclCommCount of the C ABI's own communicator
        (lisec_comm_count) when the exchange runs on it; with the exchange on torch.distributed, its world size under
        the nccl backend and 0 under gloo."""
        if self.comm is not None:
            return self._comm_ranks
        return self.world if dist.get_backend() == "nccl" else 0

    def gather_floats(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.on_gpu else "cpu")
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def shard(self, items):
        """Whole samples are dealt round-robin: rank r takes items r, r+world, ... (equal counts; the tail
        that does not fill a full round is dropped, so every rank runs the same number of steps)."""
        n = len(items) // self.world * self.world
        return list(items[self.rank:n:self.world])

    def broadcast_(self, tensor, src=0):
        dist.broadcast(tensor, src=src)
        return tensor

    def average_(self, tensor):
        """In-place mean over ranks (sum all-reduce, then * 1/world)."""
        if self.world == 1:
            return tensor
        if self.comm is not None and tensor.is_cuda and tensor.dtype == torch.float32 and tensor.numel() % 4 == 0:
            from . import _lib
            _lib.check(_lib.load().lisec_allreduce_grads(self.comm, _lib.ptr(tensor), tensor.numel(), 0,
                                                         _lib.current_stream()))    # 0: mean over the communicator
            return tensor
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        if self.on_gpu:
            from . import ops
            ops.scale_(tensor, 1.0 / self.world)
        else:
            tensor.mul_(1.0 / self.world)
        return tensor

    def bucketed(self):
        """Gradient averaging in two buckets for LisecNet.train_step: start_tail() launches the all-reduce of the
        RPN/head gradients asynchronously (from the stream that produced them), finish() reduces the small head of
        the buffer, waits for the tail and scales everything by 1/world."""
        return _BucketedAverage(self)

    def max_float(self, x):
        t = torch.tensor([x], dtype=torch.float64, device=self.device if self.on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        if self.on_gpu and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[self.device.index])
        else:
            dist.barrier()

    def close(self):
        if self.comm is not None:
            from . import _lib
            torch.cuda.synchronize(self.device)
            _lib.load().lisec_comm_destroy(self.comm)
            self.comm = None
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()


class _BucketedAverage:
    def __init__(self, dp):
        self.dp, self.work, self.lo = dp, None, 0
        # LISEC_FORCE_DP=1: issue the collectives even with a single rank (rehearsal of the RCCL path)
        self.active = dp.world > 1 or os.environ.get("LISEC_FORCE_DP") == "1"

    def __call__(self, grad):                      # plain single-bucket fallback
        return self.dp.average_(grad)

    def _enqueue(self, grad, lo, hi, after):
        """lisec_allreduce_grads of grad[lo:hi] on the communicator's own stream, after everything enqueued so far on
        the stream handle `after` (one stream per communicator keeps RCCL's issue order trivially identical on
        every rank; the later kernels of the producing stream do not queue behind the collective)."""
        from . import _lib
        dp = self.dp
        ev = dp._ev_tail if lo > 0 else dp._ev_head
        ev.record(after)
        ev.wait(dp.comm_stream.cuda_stream)
        _lib.check(_lib.load().lisec_allreduce_grads(dp.comm, grad.data_ptr() + 4 * lo, hi - lo, 0,
                                                     dp.comm_stream.cuda_stream))           # 0: mean over the communicator

    def start_tail(self, grad, lo, hi):
        if not self.active:
            return
        self.lo = lo
        if self.dp.comm is not None:
            from . import _lib
            self._enqueue(grad, lo, hi, _lib.current_stream())
            self.work = "rccl"
            return
        # torch.distributed carries the exchange: as a host call of the library, so that a recording step plan replays it at
        # this place of the sequence (the stream that is current NOW -- the second stream of the backward pass -- is bound
        # into the call; gloo reduces on the host and synchronises that stream itself)
        from . import _lib
        stream = torch.cuda.current_stream() if grad.is_cuda else None
        sl = grad[lo:hi]

        def issue():
            if stream is not None:
                with torch.cuda.stream(stream):
                    self.work = dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True)
            else:
                self.work = dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True)
        if grad.is_cuda:
            _lib.host_call(issue)
        else:
            issue()

    def finish(self, grad):
        if not self.active:
            return grad
        if self.dp.comm is not None:
            from . import _lib
            lo = self.lo if self.work is not None else grad.numel()
            self._enqueue(grad, 0, lo, _lib.current_stream())       # the head (or everything, if no tail went out)
            self.work = None
            done = self.dp._ev_done
            done.record(self.dp.comm_stream.cuda_stream)
            done.wait(_lib.current_stream())                        # the optimizer reads the averaged gradient
            return grad
        if not grad.is_cuda:
            if self.work is None:
                return self.dp.average_(grad)
            dist.all_reduce(grad[:self.lo], op=dist.ReduceOp.SUM)
            self.work.wait()
            self.work = None
            grad.mul_(1.0 / self.dp.world)
            return grad
        from . import _lib, ops
        stream = torch.cuda.current_stream()
        had_tail = self.work is not None
        head = grad[:self.lo] if had_tail else grad

        def issue():
            with torch.cuda.stream(stream):
                dist.all_reduce(head, op=dist.ReduceOp.SUM)
                if self.work is not None:
                    self.work.wait()               # the current stream waits for the tail bucket
                    self.work = None
        _lib.host_call(issue)
        ops.scale_(grad, 1.0 / self.dp.world)      # (a launch of the library: recorded by itself)
        return grad
