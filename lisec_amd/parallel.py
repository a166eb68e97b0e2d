"""Data parallelism over whole samples: one process per GPU, RCCL all-reduce of the flat gradient.

The reference has no distributed code (single process, CPU).  Its fit(batch_size=1) semantics are kept
per replica (BatchNormalization statistics are per-sample, never synchronised); the only exchange per
step is ONE all-reduce of the contiguous 6.49 M-float gradient buffer (26 MB) over xGMI, followed by
identical SGD-Nesterov updates on every rank.  backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests of this host logic.
"""
import os

import torch
import torch.distributed as dist


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
        if self._own_group and dist.is_initialized():
            dist.destroy_process_group()


class _BucketedAverage:
    def __init__(self, dp):
        self.dp, self.work, self.lo = dp, None, 0
        # LISEC_FORCE_DP=1: issue the collectives even with a single rank (rehearsal of the RCCL path)
        self.active = dp.world > 1 or os.environ.get("LISEC_FORCE_DP") == "1"

    def __call__(self, grad):                      # plain single-bucket fallback
        return self.dp.average_(grad)

    def start_tail(self, grad, lo, hi):
        if not self.active:
            return
        self.lo = lo
        self.work = dist.all_reduce(grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self, grad):
        if not self.active:
            return grad
        if self.work is None:
            return self.dp.average_(grad)
        dist.all_reduce(grad[:self.lo], op=dist.ReduceOp.SUM)
        self.work.wait()                           # the current stream waits for the tail bucket
        self.work = None
        if self.dp.on_gpu:
            from . import ops
            ops.scale_(grad, 1.0 / self.dp.world)
        else:
            grad.mul_(1.0 / self.dp.world)
        return grad
