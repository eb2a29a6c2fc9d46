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


class _Bucket:
    """One flat gradient buffer and the views its parameters' gradients live in."""

    def __init__(self, params):
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def gather_in_place(self):
        """Every parameter's gradient INTO its view, .grad re-bound to the view.  A gradient the producing kernel already
        wrote there (doda_amd's conv weight gradients and BatchNorm gamma / beta, through the extension's gradient
        homes) costs nothing; a stray one (another producer, an accumulated second pass that left its home) is copied;
        a missing one counts as zeros, so every rank sends the same message whatever its batch touched."""
        if self.params and self.params[0].is_cuda:
            try:
                from ._ext import ext as _ext
            except Exception:
                _ext = None
            if _ext is not None and hasattr(_ext, "grads_into_views"):
                return int(_ext.grads_into_views(self.params, self.views))   # (the same pass without the interpreter)
        moved = 0
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                v.copy_(g)
                if g.is_cuda:      # (the copy may run on a side stream: the block must not go back to its own stream's pool before it)
                    g.record_stream(torch.cuda.current_stream(g.device))
            else:
                continue
            p.grad = v
            moved += 1
        return moved


class GradAllReduce:
    """Gradient averaging after the backward pass.  Replaces DistributedDataParallel
    (tool/train.py:360-361) when weight gradients are deferred to the end of backward
    (doda_amd.spconv.functional.set_deferred_wgrad): those never pass through the AccumulateGrad hooks
    DDP listens on.

    * construction: parameters AND buffers (BatchNorm / DSNorm running statistics) are broadcast from
      rank 0, as DDP does at wrap time; `sync_buffers()` repeats the buffer broadcast on demand (DDP's
      broadcast_buffers=True does it every forward; call it before evaluation / checkpointing — the
      running statistics of rank-local batches otherwise drift apart, which only matters there);
    * gradients live IN the buckets (round 4; DDP's gradient_as_bucket_view): the FIXED parameter list is cut into
      persistent flat buffers of ~`bucket_mb` MB and every parameter gets a view of its bucket as its gradient's home
      (extension: set_grad_home) — the deferred weight-gradient launch and the BatchNorm backward kernels write there
      directly and .grad is an alias of the view, so reduce() issues the all-reduce on the flat buffer in place: no
      torch.cat before it, no split / copy-back after it (the r3 form moved all 30 MB twice per step);
    * reduce(): bucket k's all-reduce (RCCL ring over xGMI, per-link bound; ReduceOp.AVG where the backend has it) is
      asynchronous.  With world size 1 everything is a no-op unless `force` (or DODA_DIST_FORCE=1) asks for the
      collectives anyway (process group of one rank)."""

    def __init__(self, module, bucket_mb=8.0, broadcast_buffers=True, overlap=True, force=None):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("DODA_DIST_FORCE", "0") == "1"
        # the collectives run when there is someone to talk to, or when forced on an initialised group of one
        self.active = dist.is_initialized() and (self.world > 1 or bool(force))
        # (overlap) conv weights of the 16- / 32-channel levels: their gradients are the LAST kernels of a step
        # (the tile / pair-list launches of the deferred flush) and 2 of the 30 MB; everything else is reduced on a
        # side stream while those kernels run.  Static rule on the weight shape: identical on every rank.
        self._split = False
        self._ext = None
        self.last_moved = 0
        narrow = []
        self.armed = False          # arm(): the NEXT backward pass is the last one before reduce()
        self.time_window = False    # measurement aid: time from the early start to the end of backward (overlap_window_ms)
        self._ev_early = self._ev_done = None
        self._early_pending = None
        self._early_mode = False
        try:
            from ._ext import ext as _ext
        except Exception:
            _ext = None
        on_gpu = bool(self.params) and self.params[0].is_cuda
        if overlap and self.active and on_gpu and _ext is not None and hasattr(_ext, "set_wgrad_split"):
            narrow = [p for p in self.params if p.dim() == 5 and p.shape[3] <= 32 and p.shape[4] <= 32]
            # (round 5) a U-Net: the LATE set is instead what the backward pass produces last — the encoder of levels 1-2
            # (input conv, blocks and strided conv of the two finest levels: 0.2 of the 30 MB); everything else is complete
            # when backward leaves level 3 and its exchange starts THERE, under the level-2 / level-1 backward kernels
            # (model.py fires `start_early` from a tensor hook; reference: DDP's bucket hooks, tool/train.py:360-361)
            late_prefixes = ("input_conv.", "unet.blocks.", "unet.conv.", "unet.u.blocks.", "unet.u.conv.")
            named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
            enc = [p for n, p in named if n.startswith(late_prefixes)]
            # OPT-IN (DODA_EARLY_ALLREDUCE=1): measured under a forced one-rank RCCL group the mid-backward flush + collective
            # issue cost the host-bound step 0.8-1.0 ms (6.8-7.0 against 5.7-6.0 ms with the end-of-backward split below),
            # more than the 0.35-0.7 ms an 8-GPU ring would expose
            if (os.environ.get("DODA_EARLY_ALLREDUCE", "0") == "1" and hasattr(_ext, "flush_wgrads_early") and enc
                    and len(enc) < len(named) and any(n.startswith("unet.u.u.") for n, _ in named)):
                narrow = enc
                self._early_mode = True
            if narrow and len(narrow) < len(self.params):
                self._split = True
                from .streams import independent_stream
                self._side = independent_stream(self.params[0].device, tag="allreduce")
                _ext.set_wgrad_split(True)
        narrow_ids = {id(p) for p in narrow} if self._split else set()
        self.buckets, self.late_buckets = [], []
        if self.active:
            self.buckets = [_Bucket(b) for b in self._cut([p for p in self.params if id(p) not in narrow_ids], bucket_mb)]
            self.late_buckets = [_Bucket(b) for b in self._cut([p for p in self.params if id(p) in narrow_ids], bucket_mb)]
            if on_gpu and _ext is not None and hasattr(_ext, "set_grad_home"):
                self._ext = _ext
                for b in self.buckets + self.late_buckets:
                    for p, v in zip(b.params, b.views):
                        _ext.set_grad_home(p, v)
            _flat_broadcast(self.params)
            if broadcast_buffers:
                self.sync_buffers()
        # SUM + scale where the backend has no averaging reduction (gloo)
        self._avg = self.active and dist.get_backend() == "nccl" and hasattr(dist.ReduceOp, "AVG")

    def close(self):
        """Forget the gradient homes THIS reducer registered (the parameters' next gradients are ordinary tensors again).
        A home that a later reducer over the same parameters has replaced is left alone."""
        if self._ext is not None:
            for b in self.buckets + self.late_buckets:
                for p, v in zip(b.params, b.views):
                    self._ext.drop_grad_home(p, v)
            self._ext = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _cut(params, bucket_mb):
        """Consecutive parameters of one dtype, ~bucket_mb MB each."""
        buckets, cur, cur_bytes, limit = [], [], 0, int(bucket_mb * (1 << 20))
        for p in params:
            if cur and p.dtype != cur[0].dtype:
                buckets.append(cur)
                cur, cur_bytes = [], 0
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

    def _start(self, buckets):
        pending = []
        for b in buckets:
            self.last_moved += b.gather_in_place()   # (diagnostic: gradients that were NOT produced in their bucket)
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            pending.append((b, dist.all_reduce(b.flat, op=op, async_op=True)))
        return pending

    def _finish(self, pending):
        for b, work in pending:
            work.wait()
            if not self._avg:
                b.flat.mul_(1.0 / self.world)

    def arm(self):
        """The next backward pass is the last before reduce(): its tensor hook (doda_amd.model) may start the exchange early.
        (tool/st.py runs two backward passes per optimizer step: only the second may.)"""
        self.armed = bool(self.active and self._split and self._early_mode)
        if self.armed:
            from . import model as _model
            _model.set_early_exchange(self._on_deep_levels_done)
        return self.armed

    def overlap_window_ms(self):
        """Main-stream time between the early start of the deep levels' all-reduce and the end of the backward pass of the
        last reduced step (time_window = True): what the exchange has to hide in.  Synchronises."""
        if self._ev_early is None or self._ev_done is None:
            return None
        self._ev_done.synchronize()
        return self._ev_early.elapsed_time(self._ev_done)

    def _on_deep_levels_done(self):
        """Fired inside backward when the gradient of level 3's input exists: every layer of levels >= 3 and the whole decoder
        have run.  Issues their queued weight gradients now (main stream) and starts the all-reduce of their buckets on the
        side stream behind that launch."""
        if not self.armed or self._early_pending is not None:
            return
        self.armed = False
        from ._ext import ext as _ext
        self.last_moved = 0
        _ext.flush_wgrads_early()
        if self.time_window:
            self._ev_early = torch.cuda.Event(enable_timing=True)
            self._ev_early.record()
        if _ext.wait_wide_wgrads(self._side.cuda_stream):
            with torch.cuda.stream(self._side):
                self._early_pending = self._start(self.buckets)

    def reduce(self):
        """Call between loss.backward() and optimizer.step()."""
        if not self.active:
            return
        self.armed = False
        if self._early_pending is not None:     # the deep levels' buckets have been travelling since mid-backward
            main = torch.cuda.current_stream()
            if self.time_window and self._ev_early is not None:
                self._ev_done = torch.cuda.Event(enable_timing=True)
                self._ev_done.record()
            early, self._early_pending = self._early_pending, None
            late = self._start(self.late_buckets)
            with torch.cuda.stream(self._side):
                self._finish(early)
            self._finish(late)
            main.wait_stream(self._side)
            from ._ext import ext as _ext
            _ext.wait_wide_wgrads(self._side.cuda_stream)   # (consume the split flush's event of this pass, if any)
            return
        self.last_moved = 0
        main = torch.cuda.current_stream() if self._split else None
        if self._split and self._ext_wait():
            # the side stream now waits for the wide layers' weight gradients only (and for everything backward
            # put on the stream before them); the narrow layers' kernels keep running on the main stream
            # (the buckets are persistent: no allocator hand-over between the streams)
            with torch.cuda.stream(self._side):
                early = self._start(self.buckets)
            late = self._start(self.late_buckets)
            with torch.cuda.stream(self._side):
                self._finish(early)
            self._finish(late)
            main.wait_stream(self._side)
            return
        self._finish(self._start(self.buckets + self.late_buckets))

    def _ext_wait(self):
        from ._ext import ext as _ext
        return _ext.wait_wide_wgrads(self._side.cuda_stream)
