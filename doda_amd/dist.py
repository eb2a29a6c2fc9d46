"""Data-parallel plumbing for the U-Net step: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) on GPUs, gloo on CPU (tests).  Scenes are independent units, so ranks share
nothing but the gradient all-reduce (reference tool/train.py:360-361: DDP) and the bench bookkeeping
below (max-over-ranks time, sum-over-ranks work)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def setup(backend=None):
    """Initialise the default process group from the launcher's environment (idempotent).
    Returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    # DODA_DIST_FORCE=1: a launcher-started single rank still joins a process group and runs the collectives
    # (a 1-rank RCCL communicator executes the same all-reduce / broadcast calls: the way to exercise the
    # RCCL streams on a one-GPU box)
    forced = os.environ.get("DODA_DIST_FORCE", "0") == "1" and "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
        if backend is None:
            backend = os.environ.get("DODA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend)
    return world, rank, local_rank


def barrier():
    if dist.is_initialized():
        dist.barrier()


def seed_for_rank(base_seed, rank):
    """Every rank draws its own scenes (weak scaling): disjoint seed ranges per rank."""
    return base_seed + 100 * rank


def reduce_step_stats(elapsed, units, device):
    """elapsed -> MAX over ranks, every entry of `units` -> SUM over ranks (whole-job aggregate)."""
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    u = torch.tensor([float(v) for v in units], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), [float(v) for v in u]


def wrap_ddp(module, local_rank=None, broadcast_buffers=True):
    """DistributedDataParallel as tool/train.py:360-361 wraps the model (defaults: buffers broadcast
    from rank 0 on every forward); gradients live in bucket views."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return module
    kw = dict(broadcast_buffers=bool(broadcast_buffers), gradient_as_bucket_view=True)
    if local_rank is not None and torch.cuda.is_available():
        kw["device_ids"] = [local_rank]
    return torch.nn.parallel.DistributedDataParallel(module, **kw)


def _flat_broadcast(tensors, src=0):
    """Broadcast a list of tensors from `src` as one flat message per dtype."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for group in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src)
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


class GradAllReduce:
    """Gradient averaging after the backward pass.  Replaces DistributedDataParallel
    (tool/train.py:360-361) when weight gradients are deferred to the end of backward
    (doda_amd.spconv.functional.set_deferred_wgrad): those never pass through the AccumulateGrad hooks
    DDP listens on.

    * construction: parameters AND buffers (BatchNorm / DSNorm running statistics) are broadcast from
      rank 0, as DDP does at wrap time; `sync_buffers()` repeats the buffer broadcast on demand (DDP's
      broadcast_buffers=True does it every forward; call it before evaluation / checkpointing — the
      running statistics of rank-local batches otherwise drift apart, which only matters there);
    * reduce(): the FIXED parameter list (a missing gradient counts as zeros, so every rank sends the
      same message sizes whatever its batch touched) is cut into buckets of ~`bucket_mb` MB; bucket k's
      all-reduce (RCCL ring over xGMI, per-link bound) is asynchronous and overlaps the flattening copy
      of bucket k+1 and the copy-back of bucket k-1.  With world size 1 everything is a no-op unless
      `force` (or DODA_DIST_FORCE=1) asks for the collectives anyway (process group of one rank)."""

    def __init__(self, module, bucket_mb=8.0, broadcast_buffers=True, overlap=True, force=None):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("DODA_DIST_FORCE", "0") == "1"
        # the collectives run when there is someone to talk to, or when forced on an initialised group of one
        self.active = dist.is_initialized() and (self.world > 1 or bool(force))
        # (overlap) conv weights of the 16- / 32-channel levels: their gradients are the LAST kernels of a step
        # (the pair-list launches of the deferred flush) and 2 of the 30 MB; everything else is reduced on a
        # side stream while those kernels run.  Static rule on the weight shape: identical on every rank.
        self._split = False
        narrow = []
        if overlap and self.active and self.params and self.params[0].is_cuda:
            try:
                from ._ext import ext as _ext
            except Exception:
                _ext = None
            if _ext is not None and hasattr(_ext, "set_wgrad_split"):
                narrow = [p for p in self.params if p.dim() == 5 and p.shape[3] <= 32 and p.shape[4] <= 32]
                if narrow and len(narrow) < len(self.params):
                    self._split, self._ext = True, _ext
                    from .streams import independent_stream
                    self._side = independent_stream(self.params[0].device, tag="allreduce")
                    _ext.set_wgrad_split(True)
        narrow_ids = {id(p) for p in narrow} if self._split else set()
        self.buckets = self._cut([p for p in self.params if id(p) not in narrow_ids], bucket_mb)
        self.late_buckets = self._cut([p for p in self.params if id(p) in narrow_ids], bucket_mb)
        if self.active:
            _flat_broadcast(self.params)
            if broadcast_buffers:
                self.sync_buffers()

    @staticmethod
    def _cut(params, bucket_mb):
        buckets, cur, cur_bytes, limit = [], [], 0, int(bucket_mb * (1 << 20))
        for p in params:
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= limit:
                buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            buckets.append(cur)
        return buckets

    def sync_buffers(self):
        """Rank 0's buffers (running statistics, batch counters) to every rank."""
        if self.active:
            bufs = [b for b in self.module.buffers() if b is not None and b.numel() > 0]
            if bufs:
                _flat_broadcast(bufs)

    @staticmethod
    def _start(buckets):
        pending = []
        for bucket in buckets:
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
            pending.append((bucket, flat, dist.all_reduce(flat, async_op=True)))
        return pending

    def _finish(self, pending):
        inv = 1.0 / self.world
        for bucket, flat, work in pending:
            work.wait()
            flat.mul_(inv)
            views = [t.view_as(p) for t, p in zip(flat.split([p.numel() for p in bucket]), bucket)]
            have = [(p.grad, v) for p, v in zip(bucket, views) if p.grad is not None]
            if have:
                torch._foreach_copy_([g for g, _ in have], [v for _, v in have])
            for p, v in zip(bucket, views):
                if p.grad is None:
                    p.grad = v.clone()

    def reduce(self):
        """Call between loss.backward() and optimizer.step()."""
        if not self.active:
            return
        main = torch.cuda.current_stream() if self._split else None
        if self._split and self._ext.wait_wide_wgrads(self._side.cuda_stream):
            # the side stream now waits for the wide layers' weight gradients only (and for everything backward
            # put on the stream before them); the narrow layers' kernels keep running on the main stream
            # (gradients stay alive until the next zero_grad, which the main stream reaches after it has
            # joined the side stream: no allocator hand-over needed)
            with torch.cuda.stream(self._side):
                early = self._start(self.buckets)
            late = self._start(self.late_buckets)
            with torch.cuda.stream(self._side):
                self._finish(early)
            self._finish(late)
            main.wait_stream(self._side)
            return
        self._finish(self._start(self.buckets + self.late_buckets))
