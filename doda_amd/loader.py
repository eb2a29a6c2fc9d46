"""Non-blocking loader behind the training entry point.

Reference: dataset/__init__.py:62-75 builds a torch DataLoader (worker processes, DistributedSampler, the dataset's
collate_fn doing concatenation + CPU voxelisation in the workers, dataset/dataset.py:121-187); tool/train.py:69-158 then
copies every batch to the GPU inside the step.  Here the same three stages run AHEAD of the step and none of them on the
issuing thread:

  1. worker processes (torch.utils.data.DataLoader, `--workers`, forkserver context: no HIP state is forked) return the
     scenes of a batch as the tuples a DODA dataset's __getitem__ returns (dataset/dataset.py:136);
  2. one feeder thread per process uploads them and collates ON THE DEVICE (doda_amd.collate: doda_voxelize_idx_assign /
     _fill instead of the workers' CPU voxelisation) and builds the batch's 13 rulebooks, all on the rulebook stream — the
     size read-backs of both block this thread only;
  3. the training loop takes a finished (batch, rulebooks) pair with one stream wait.

Data: there is no dataset in this environment.  `SyntheticScenes` stands in for dataset/scannet.py: a pool of procedural
base scenes (doda_amd.scene, generated once into a cache directory, as the reference preprocesses its scans into
SharedArray, dataset/scannet.py:22-28) and, per sample, what the reference's loader does per sample — a rigid augmentation
(z-rotation, x-flip, jitter; dataset/augmentor_utils.py:85-104) and the integer voxel coordinates `xyz * voxel_scale - min`
(dataset/scannet.py:76-78)."""
import atexit
import os
import queue
import shutil
import tempfile
import threading

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ base scenes
def _make_base(job):
    """(worker of prepare_cache) one base scene -> cache file."""
    path, seed, voxels, voxel_scale = job
    if os.path.exists(path):
        return path
    from .scene import make_scene   # (imported here: the pool's workers do not need it before)
    _, xyz_mid, labels = make_scene(seed, voxels, voxel_scale)
    tmp = path + ".tmp.%d.npz" % os.getpid()
    np.savez(tmp, xyz_mid=xyz_mid.astype(np.float32), labels=labels.astype(np.int64))
    os.replace(tmp, path)
    return path


def prepare_cache(n_base, voxels, voxel_scale, seed0, cache_dir=None, procs=None):
    """Generate (or find) the pool of base scenes; returns (cache_dir, [paths]).  ~2 s of numpy per 150 k-voxel scene,
    spread over `procs` processes; files are keyed by (seed, voxels, scale), so ranks and later runs share them."""
    if cache_dir is None:
        root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        cache_dir = os.path.join(root, "doda_amd_scenes_%d" % os.getuid())
    os.makedirs(cache_dir, exist_ok=True)
    jobs = [(os.path.join(cache_dir, "scene_s%d_v%d_x%d.npz" % (seed0 + k, voxels, voxel_scale)), seed0 + k, voxels, voxel_scale)
            for k in range(n_base)]
    todo = [j for j in jobs if not os.path.exists(j[0])]
    if todo:
        procs = min(len(todo), procs or min(16, os.cpu_count() or 1))
        if procs > 1:
            import multiprocessing as mp
            with mp.get_context("forkserver").Pool(procs) as pool:
                pool.map(_make_base, todo)
        else:
            for j in todo:
                _make_base(j)
    return cache_dir, [j[0] for j in jobs]


def remove_cache(cache_dir):
    shutil.rmtree(cache_dir, ignore_errors=True)


class SyntheticScenes(torch.utils.data.Dataset):
    """item i of epoch e -> (xyz int64 [n,3] voxel coordinates, xyz_mid float32 [n,3], labels int64 [n], id): base scene
    i % n_base under a seeded rigid augmentation.  Picklable (paths + numbers only): workers open the cache files."""

    def __init__(self, paths, length, voxel_scale, seed, augment=True):
        self.paths, self.length, self.voxel_scale, self.seed, self.augment = list(paths), int(length), int(voxel_scale), int(seed), augment
        self._mem = {}

    def __len__(self):
        return self.length

    def _base(self, k):
        v = self._mem.get(k)
        if v is None:
            with np.load(self.paths[k]) as f:
                v = self._mem[k] = (np.ascontiguousarray(f["xyz_mid"].T, dtype=np.float32), f["labels"].astype(np.int32))
        return v

    def __getitem__(self, i):
        # (i = epoch * length + index: the sampler hands out global indices, so persistent workers need no per-epoch state)
        xt, labels = self._base((i % self.length) % len(self.paths))      # xt: float32 [3, n] (columns contiguous)
        n = xt.shape[1]
        m = np.empty((n, 3), dtype=np.float32)
        if self.augment:
            # (column arithmetic: numpy's [n, 3] @ [3, 3] took 48 ms for 200 k points, its axis-0 minimum 4 ms)
            rng = np.random.default_rng((self.seed * 1000003 + i) & 0x7fffffff)
            th = rng.uniform(0.0, 2.0 * np.pi)
            c, s = np.float32(np.cos(th)), np.float32(np.sin(th))
            flip = np.float32(-1.0 if rng.random() < 0.5 else 1.0)
            jit = rng.random((3, n), dtype=np.float32)
            jit -= np.float32(0.5)
            jit *= np.float32(0.01)
            x = xt[0] * flip
            np.add(x * c - xt[1] * s, jit[0], out=m[:, 0])
            np.add(x * s + xt[1] * c, jit[1], out=m[:, 1])
            np.add(xt[2], jit[2], out=m[:, 2])
        else:
            m[:, 0], m[:, 1], m[:, 2] = xt[0], xt[1], xt[2]
        q = np.empty((n, 3), dtype=np.int32)
        for k in range(3):
            col = m[:, k] * np.float32(self.voxel_scale)
            col -= col.min()
            q[:, k] = col          # (truncation, as the reference's astype: dataset/scannet.py:78)
        # tensors (the DataLoader hands tensors over in shared memory; numpy arrays would be pickled through a pipe) in the
        # narrowest types that hold the values: 28 instead of 56 bytes per point cross the process boundary and PCIe —
        # doda_amd.collate widens them to the reference's int64 / float32 ON THE DEVICE
        return torch.from_numpy(q), torch.from_numpy(m), torch.from_numpy(labels), int(i)


def _identity(items):
    return items


def host_collate(items, full_scale0=128):
    """collate_fn run IN THE WORKERS (as the reference's, dataset/dataset.py:121-187: concatenation, batch-index column,
    offsets, spatial_shape) — minus the CPU voxelisation, which the device does (doda_amd.collate.collate_device_concat).
    One int32 [N, 4], one float32 [N, 3] and one int32 [N] tensor per batch cross the process boundary and PCIe: three
    uploads per batch instead of twelve, and no `max` read-back on the feeder's stream."""
    locs, mids, labs, ids, offsets = [], [], [], [], [0]
    top = np.zeros(3, dtype=np.int64)
    for b, (xyz, mid, lab, idx) in enumerate(items):
        n = xyz.shape[0]
        offsets.append(offsets[-1] + n)
        lb = torch.empty((n, 4), dtype=torch.int32)
        lb[:, 0] = b
        lb[:, 1:] = xyz
        locs.append(lb)
        mids.append(mid)
        labs.append(lab)
        ids.append(idx)
        if n:
            top = np.maximum(top, xyz.max(0)[0].numpy().astype(np.int64) + 1)
    return {"locs32": torch.cat(locs, 0), "locs_float": torch.cat(mids, 0), "labels32": torch.cat(labs, 0),
            "offsets": torch.tensor(offsets, dtype=torch.int32), "spatial_shape": np.clip(top, full_scale0, None), "id": ids}


class EpochSampler(torch.utils.data.Sampler):
    """DistributedSampler semantics over global indices: one seeded permutation per epoch, rank r takes positions r,
    r + world, ...; the tail that does not fill a batch is dropped.  set_epoch() before each epoch's iterator."""

    def __init__(self, length, batch_size, rank, world, shuffle=True, seed=0):
        self.length, self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch = length, batch_size, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _mine(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.length, generator=g).tolist()
        else:
            order = list(range(self.length))
        mine = order[self.rank::self.world]
        return mine[:len(mine) // self.bs * self.bs]

    def __len__(self):
        return len(self._mine())

    def __iter__(self):
        base = self.epoch * self.length if self.shuffle else 0
        return iter([base + i for i in self._mine()])


def host_loader(dataset, batch_size, rank, world, workers, shuffle=True, seed=0, full_scale0=128):
    """(DataLoader of lists of `batch_size` dataset items for this rank, its EpochSampler) — reference
    dataset/__init__.py:62-75 with the collate left to the device."""
    sampler = EpochSampler(len(dataset), batch_size, rank, world, shuffle, seed)
    kw = {}
    if workers > 0:
        kw = dict(num_workers=workers, multiprocessing_context="forkserver", prefetch_factor=4, persistent_workers=True)
    import functools
    # (full_scale0: cfg DATA_PROCESSOR.full_scale[0] — the clip of spatial_shape, reference dataset/dataset.py:176 — as DeviceScenes
    # and collate_device take it: the two loaders must agree for any config)
    collate = host_collate if int(full_scale0) == 128 else functools.partial(host_collate, full_scale0=int(full_scale0))
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, collate_fn=collate, drop_last=True,
                                       pin_memory=torch.cuda.is_available(), **kw), sampler


# ------------------------------------------------------------------------------------------------ device side
class DeviceFeeder:
    """Feeder thread: host batches -> device batches + rulebooks, `depth` batches ahead of the consumer.

    Iterating yields (batch dictionary on `device`, (indices int32, indice_dict) | None), already handed over to the
    consumer's current stream.  Exceptions of the thread (or of a worker) are re-raised in the consumer."""

    _END = object()

    def __init__(self, host_iter, device, prefetcher=None, with_pairs=False, with_tiles=None, voxel_mode=4,
                 full_scale=(128, 512), depth=2, voxel_order="auto"):
        self.device, self.prefetcher = torch.device(device), prefetcher
        # numbering of the voxels (collate.reorder_voxels): "first" = the reference's, "morton", or "auto" = decided from the
        # first batch's tile overflow (collate.choose_voxel_order) and kept
        self.voxel_order = voxel_order
        self.host_iter = host_iter
        self.kw = dict(voxel_mode=voxel_mode, full_scale=full_scale)
        self.with_pairs, self.with_tiles = with_pairs, with_tiles
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = False
        self.wait_host_s = self.collate_s = 0.0     # where the feeder thread's time went (waiting for the workers / on the device)
        self.batches = 0
        from .streams import independent_stream
        self.stream = independent_stream(self.device, tag="loader")   # (the rulebook builds keep their own stream and thread)
        self.thread = threading.Thread(target=self._run, name="doda-feeder", daemon=True)
        self.thread.start()

    def _run(self):
        from .collate import choose_voxel_order, collate_device, collate_device_concat, reorder_voxels
        try:
            import time
            torch.cuda.set_device(self.device)
            t_prev = time.perf_counter()
            it = iter(self.host_iter)
            while not self._stop:
                with torch.cuda.stream(self.stream):   # (a device-resident source makes its batch with kernels: on this stream too)
                    items = next(it, self._END)
                if items is self._END:
                    break
                t_got = time.perf_counter()
                self.wait_host_s += t_got - t_prev
                with torch.cuda.stream(self.stream):
                    if isinstance(items, dict):     # concatenated in the worker (host_collate)
                        batch = collate_device_concat(items, self.device, voxel_mode=self.kw["voxel_mode"])
                    else:
                        batch = collate_device(items, self.device, **self.kw)
                    if self.voxel_order == "auto" and batch["voxel_locs"].shape[0] > 0:
                        self.voxel_order = choose_voxel_order(batch, self.device)
                    if self.voxel_order == "morton":
                        batch = reorder_voxels(batch, "morton")
                    fut = None
                    if self.prefetcher is not None and batch["voxel_locs"].shape[0] > 0:
                        # second pipeline stage: the rulebook thread builds this batch's 13 rulebooks (behind an event recorded
                        # here, on the loader stream) while this thread goes back for the next batch
                        fut = self.prefetcher.submit(batch, self.with_pairs, self.with_tiles, now=True)
                    done = torch.cuda.Event()
                    done.record(self.stream)
                self.collate_s += time.perf_counter() - t_got
                self.batches += 1
                self._put((batch, fut, done))
                t_prev = time.perf_counter()
            self._put(self._END)
        except BaseException as e:   # noqa: BLE001  (handed to the consumer)
            self._put(e)

    def _put(self, item):
        while not self._stop:
            try:
                self.q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is self._END:
            self.thread.join()
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        batch, fut, done = item
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        for v in batch.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(main)
        pyramid = self.prefetcher.take(fut, self.device) if fut is not None else None
        return batch, pyramid

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=5.0)


class DeviceScenes:
    """The dataset RESIDENT IN HBM (MI355X-first: 288 GB per GPU hold any of DODA's datasets — ScanNet's 1 201 training
    scans are ~5 GB at 28 bytes per point — so nothing has to cross PCIe per step): the base scenes are uploaded once and a
    batch is made by a dozen small kernels on the loader stream — per-scene z-rotation / x-flip / jitter / integer voxel
    coordinates (what the reference does per sample on the CPU: dataset/scannet.py:76-78, augmentor_utils.py:85-104) and the
    concatenation of dataset/dataset.py:121-187.  Iterating yields the dictionary host_collate() makes, on the device.
    Same sampling order as EpochSampler; the augmentation draws come from a seeded DEVICE generator (not the worker path's
    numpy streams: the two loaders produce different, equally distributed batches)."""

    def __init__(self, paths, length, voxel_scale, seed, batch_size, rank, world, device, augment=True, shuffle=True,
                 full_scale0=128):
        self.device = torch.device(device)
        self.length, self.voxel_scale, self.seed, self.bs, self.augment = int(length), float(voxel_scale), int(seed), batch_size, augment
        self.full_scale0 = full_scale0
        self.sampler = EpochSampler(length, batch_size, rank, world, shuffle, seed)
        self.xyz, self.lab = [], []
        for p in paths:
            with np.load(p) as f:
                self.xyz.append(torch.from_numpy(np.ascontiguousarray(f["xyz_mid"], dtype=np.float32)).to(self.device))
                self.lab.append(torch.from_numpy(f["labels"].astype(np.int32)).to(self.device))
        self.gen = torch.Generator(device=self.device)

    def set_epoch(self, epoch):
        self.sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.sampler) // self.bs

    def __iter__(self):
        idx = list(self.sampler)
        for b0 in range(0, len(idx) - self.bs + 1, self.bs):
            yield self._batch(idx[b0:b0 + self.bs])

    @torch.no_grad()
    def _batch(self, ids):
        dev, bs = self.device, len(ids)
        base = [(i % self.length) % len(self.xyz) for i in ids]
        sizes = [self.xyz[k].shape[0] for k in base]
        offsets = [0]
        for n in sizes:
            offsets.append(offsets[-1] + n)
        x = torch.cat([self.xyz[k] for k in base], 0)                       # [N, 3] float32
        labels = torch.cat([self.lab[k] for k in base], 0)
        bidx = torch.repeat_interleave(torch.arange(bs, device=dev), torch.tensor(sizes, device=dev), output_size=offsets[-1])
        if self.augment:
            # per-scene rigid transform from host-side seeded draws (tiny), jitter from the device generator
            rng = np.random.default_rng((self.seed * 1000003 + ids[0]) & 0x7fffffff)
            th = rng.uniform(0.0, 2.0 * np.pi, bs)
            flip = np.where(rng.random(bs) < 0.5, -1.0, 1.0)
            # (column arithmetic with the four coefficients gathered per point: torch.einsum over an [N, 3, 3] gather became
            # 600 k batched 1x3x3 products — 60 ms per batch)
            coef = np.stack((np.cos(th) * flip, np.sin(th) * flip, -np.sin(th), np.cos(th)), 1).astype(np.float32)
            cf = torch.from_numpy(coef).to(dev, non_blocking=True)[bidx]       # [N, 4]
            m = torch.stack((x[:, 0] * cf[:, 0] + x[:, 1] * cf[:, 2], x[:, 0] * cf[:, 1] + x[:, 1] * cf[:, 3], x[:, 2]), 1)
            self.gen.manual_seed((self.seed * 7919 + ids[0]) & 0x7fffffff)
            m += (torch.rand(m.shape, device=dev, generator=self.gen) - 0.5) * 0.01
        else:
            m = x.clone()
        q = m * self.voxel_scale
        # (per-scene minimum as one reduction per scene: a scatter_reduce of 1.8 M values onto 12 addresses is 50 ms of atomics)
        lo = torch.stack([q[offsets[b]:offsets[b + 1]].amin(0) for b in range(bs)])
        q = (q - lo[bidx]).to(torch.int32)
        top = (q.max(0)[0] + 1).cpu().numpy().astype(np.int64)              # (the loader thread's read-back)
        locs32 = torch.cat((bidx.to(torch.int32)[:, None], q), 1)
        return {"locs32": locs32, "locs_float": m, "labels32": labels, "offsets": torch.tensor(offsets, dtype=torch.int32),
                "spatial_shape": np.clip(top, self.full_scale0, None), "id": [int(i) for i in ids]}


_OWN_CACHES = []      # cache directories this PROCESS created privately (DODA_PRIVATE_SCENES=1): the only ones it may delete


def synthetic_dataset(cfg, args, split):
    """The synthetic stand-in for the reference's dataset objects: `--synthetic_scenes` items per epoch over a pool of
    min(--synthetic_base, items) base scenes per split.

    The default cache (/dev/shm/doda_amd_scenes_<uid>) is SHARED by every job of the user on the node — files are keyed by
    (seed, voxels, scale) and written atomically, so concurrent jobs and later runs reuse them — and is therefore never deleted
    by a job (an exiting job used to remove it under a running one; `python -m doda_amd.loader --clean` or
    remove_cache() empties it).  DODA_PRIVATE_SCENES=1: a per-process directory instead, removed at exit by local rank 0."""
    base_seed = {"train": 1000, "target": 501000, "val": 901000}[split]
    n_base = max(1, min(int(getattr(args, "synthetic_base", 16)), int(args.synthetic_scenes)))
    voxel_scale = cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_scale
    cache = getattr(args, "scene_cache", None)
    if cache is None and os.environ.get("DODA_PRIVATE_SCENES") == "1":
        if not _OWN_CACHES:
            root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            # (one directory per job: every rank of a torchrun job sees the same MASTER_PORT; a single process its own pid)
            _OWN_CACHES.append(os.path.join(root, "doda_amd_scenes_%d_%s" % (os.getuid(), os.environ.get("MASTER_PORT", os.getpid()))))
        cache = _OWN_CACHES[0]
    cache_dir, paths = prepare_cache(n_base, args.synthetic_voxels, voxel_scale, base_seed, cache)
    return SyntheticScenes(paths, args.synthetic_scenes, voxel_scale, seed=base_seed, augment=split != "val")


@atexit.register
def _cleanup():   # (only a private per-job cache, by the node's first rank; the shared default cache and explicit ones are left alone)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        for d in _OWN_CACHES:
            remove_cache(d)


if __name__ == "__main__":   # python -m doda_amd.loader --clean : empty the shared default scene cache of this user on this node
    import sys
    if "--clean" in sys.argv:
        root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        remove_cache(os.path.join(root, "doda_amd_scenes_%d" % os.getuid()))
