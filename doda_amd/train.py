"""Training entry point with the reference's command line, YAML configs and checkpoint format
(tool/train.py:29-62 arguments, :69-158 train_epoch, :235-268 epoch loop / checkpoints, :271-361 main;
tool/st.py:100-200 for the two-pass self-training step) on top of the MI355X-native hot path.

    python -m doda_amd.train --cfg_file cfgs/scannet/spconv.yaml [--batch_size 4] [--epochs N]
           [--resume ckpt | --weight ckpt] [--launcher pytorch] [--set KEY VALUE ...]
           [--synthetic_scenes 64] [--dtype bf16] [--self_train]

What is the same: every reference flag, `_BASE_CONFIG_` / `--set` config handling (doda_amd.config),
SGD / Adam / AdamW + step / poly / cos learning-rate schedules (util/common_utils.py:154-215), DSNorm
conversion and per-domain switching (tool/train.py:89-90,332; tool/st.py:137-164), the checkpoint
dictionary `{epoch, state_dict, optimizer, commit_id, metric}` and its `module.` stripping
(util/model_utils.py:22-94), auto-resume from the newest `train_epoch_*.pth`, best-mIoU checkpoint.

What is different on purpose (SURVEY §7 "harness overheads that cap DDP scaling"): no
`torch.cuda.empty_cache()` per iteration (tool/train.py:82), no `.item()` / CPU `histc` round trip per
iteration (tool/train.py:107-118, util/common_utils.py:242-246) — loss and the intersection / union /
target histograms accumulate ON THE DEVICE and are read back every `--print_freq` iterations —, batches
are collated on the GPU (doda_amd.collate), the rulebooks of the next batch are built during the
current step (PyramidPrefetcher), gradients are averaged by doda_amd.dist.GradAllReduce.

Data: there is no dataset in this environment, so the loader here draws seeded synthetic ScanNet-shaped
scenes (doda_amd.scene); `make_loader` is the one function to replace for real data — it must yield
the collate dictionary of reference dataset/dataset.py:121-187."""
import argparse
import glob
import math
import os
import subprocess
import time
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

from .config import Config, cfg_from_list, cfg_from_yaml_file


# ------------------------------------------------------------------------------------------------ CLI
def build_parser():
    """tool/train.py:29-52, flag for flag; the last group is this harness's own."""
    p = argparse.ArgumentParser(description="arg parser")
    p.add_argument("--cfg_file", type=str, default=None, help="specify the config for training")
    p.add_argument("--batch_size", type=int, default=None, required=False, help="batch size for training")
    p.add_argument("--epochs", type=int, default=None, required=False, help="number of epochs to train for")
    p.add_argument("--workers", type=int, default=4, help="number of workers for dataloader")
    p.add_argument("--extra_tag", type=str, default="default", help="extra tag for this experiment")
    p.add_argument("--start_epoch", type=int, default=0)
    p.add_argument("--resume", type=str, default=None, help="checkpoint to start from")
    p.add_argument("--weight", type=str, default=None, help="pretrained_model")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none")
    p.add_argument("--tcp_port", type=int, default=18867, help="tcp port for distrbuted training")
    p.add_argument("--sync_bn", action="store_true", default=False, help="whether to use sync bn")
    p.add_argument("--reserve_old_ckpt", action="store_true", default=False)
    p.add_argument("--manual_seed", type=int, default=None)
    p.add_argument("--ckpt_save_freq", type=int, default=1, help="number of training epochs")
    p.add_argument("--print_freq", type=int, default=5, help="printing log frequency")
    p.add_argument("--local_rank", type=int, default=0, help="local rank for distributed training")
    p.add_argument("--max_ckpt_save_num", type=int, default=30, help="max number of saved checkpoint")
    p.add_argument("--pretrain_not_strict", action="store_true", default=False,
                   help="(the reference reads this attribute at tool/train.py:344 without defining it)")
    p.add_argument("--pin_memory", action="store_true", default=False)
    p.add_argument("--synthetic_scenes", type=int, default=32, help="scenes per epoch of the synthetic loader")
    p.add_argument("--synthetic_voxels", type=int, default=150000, help="active voxels per synthetic scene")
    p.add_argument("--synthetic_base", type=int, default=16, help="procedural base scenes per split (samples = augmented base scenes)")
    p.add_argument("--scene_cache", type=str, default=None, help="directory of the base scenes (default: /dev/shm/doda_amd_scenes_<uid>)")
    p.add_argument("--host_loader", action="store_true", default=False,
                   help="DataLoader worker processes + uploads per batch (reference dataset/__init__.py:62-75) instead of the "
                        "dataset resident in HBM with the augmentation on the device (doda_amd.loader.DeviceScenes)")
    p.add_argument("--inline_loader", action="store_true", default=False,
                   help="generate and collate every batch in the training loop's own thread (the pre-round-5 loader)")
    p.add_argument("--voxel_order", choices=["auto", "first", "morton"], default="auto",
                   help="numbering of a batch's voxels (doda_amd.collate.reorder_voxels): the reference's first-appearance order, a "
                        "Z-order renumbering, or decided from the first batch's tile overflow (per-point outputs do not depend on it)")
    p.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="feature storage dtype")
    p.add_argument("--self_train", action="store_true", default=False,
                   help="tool/st.py step: a source pass and a target pass per optimizer step")
    p.add_argument("--output_root", type=str, default=None, help="default: <cfg root>/output")
    p.add_argument("--max_iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    p.add_argument("--timing_warmup", type=int, default=10, help="iterations of an epoch before its steady-state clock starts")
    p.add_argument("--timing_json", type=str, default=None, help="write {iterations, seconds, ms_per_iter, voxels_per_s} here at exit")
    p.add_argument("--curve_json", type=str, default=None,
                   help="write the run's curve here at exit: per epoch the mean training loss and, when evaluated, the held-out "
                        "loss / mIoU / per-class IoU (tools/trajectory.sh: bf16 against fp32 on the same data and seed)")
    p.add_argument("--set", dest="set_cfgs", default=None, nargs=argparse.REMAINDER,
                   help="set extra config keys if needed")
    return p


def parse_config(argv=None):
    args = build_parser().parse_args(argv)
    cfg = Config()
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    cfg.EXP_GROUP_PATH = "/".join(args.cfg_file.split("/")[1:-1])  # remove 'cfgs' and 'xxxx.yaml'
    cfg.LOCAL_RANK = 0
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    return args, cfg


# ------------------------------------------------------------------------------ optimizer / LR / checkpoints
def step_learning_rate(optimizer, base_lr, epoch, step_epoch, multiplier=0.1, clip=1e-6):
    lr = max(base_lr * (multiplier ** (epoch // step_epoch)), clip)
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


def poly_learning_rate(optimizer, base_lr, curr_iter, max_iter, power=0.9):
    lr = base_lr * (1 - float(curr_iter) / max_iter) ** power
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


def cos_learning_rate(optimizer, base_lr, curr_iter, max_iter, warm_iter, hold_base_iter):
    lr = 0.5 * base_lr * (1 + math.cos(math.pi * (curr_iter - warm_iter - hold_base_iter) /
                                       float(max_iter - warm_iter - hold_base_iter)))
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


def adjust_lr(optim_cfg, optimizer, scheduler, total_epochs, total_iters_per_epoch, epoch, it):
    """util/common_utils.py:176-193 (note the reference's `epoch - 1` in the step schedule)."""
    if optim_cfg.lr_decay == "step":
        return step_learning_rate(optimizer, optim_cfg.base_lr, epoch - 1, optim_cfg.step_epoch, optim_cfg.multiplier)
    if optim_cfg.lr_decay == "poly":
        return poly_learning_rate(optimizer, optim_cfg.base_lr, epoch * total_iters_per_epoch + it + 1,
                                  total_iters_per_epoch * total_epochs)
    if optim_cfg.lr_decay == "cos":
        return cos_learning_rate(optimizer, optim_cfg.base_lr, epoch * total_iters_per_epoch + it + 1,
                                 total_iters_per_epoch * total_epochs, 0, 0)
    if optim_cfg.lr_decay == "adam_onecycle":
        scheduler.step(epoch * total_iters_per_epoch + it + 1)
        return optimizer.param_groups[0]["lr"]
    if optim_cfg.lr_decay in ("multistep",):
        return optimizer.param_groups[0]["lr"]
    raise NotImplementedError(optim_cfg.lr_decay)


def build_optimizer(optim_cfg, model, fused=None):
    """util/common_utils.py:196-215.  fused: use the fused multi-tensor kernels on the GPU (one launch
    instead of one per parameter group member; the packed conv weights follow it — doda_amd.spconv.conv)."""
    params = [p for p in model.parameters() if p.requires_grad]
    kind = optim_cfg.get("optim", "sgd")
    kw = {}
    if fused is None:
        fused = bool(params) and params[0].is_cuda
    if fused:
        kw["fused"] = True
    if kind == "sgd":
        if fused and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            from .optim import FusedSGD   # torch.optim.SGD, update of all tensors in one native launch
            return FusedSGD(params, lr=optim_cfg.base_lr, momentum=optim_cfg.momentum,
                            weight_decay=optim_cfg.weight_decay)
        return torch.optim.SGD(params, lr=optim_cfg.base_lr, momentum=optim_cfg.momentum,
                               weight_decay=optim_cfg.weight_decay, **kw)
    if kind == "adam":
        return torch.optim.Adam(params, lr=optim_cfg.base_lr, **kw)
    if kind == "adamw":
        return torch.optim.AdamW(params, lr=optim_cfg.base_lr, **kw)
    raise NotImplementedError(kind)


def get_git_commit_id():
    if not os.path.exists(".git"):
        return "0000000"
    out = subprocess.run(["git", "rev-parse", "HEAD"], stdout=subprocess.PIPE)
    return out.stdout.decode("utf-8")[:7]


def save_params(filename, model, optimizer, epoch_log, metric=None):
    """util/model_utils.py:87-94: the reference's checkpoint dictionary, model state on the CPU."""
    net = model.module if hasattr(model, "module") else model
    state = OrderedDict((k, v.cpu()) for k, v in net.state_dict().items())
    torch.save({"epoch": epoch_log, "state_dict": state, "optimizer": optimizer.state_dict(),
                "commit_id": get_git_commit_id(), "metric": metric}, filename)


def get_ckpt(path, to_cpu=True):
    ckpt = torch.load(path, map_location=torch.device("cpu") if to_cpu else None, weights_only=False)
    if "state_dict" in ckpt:   # checkpoints written from a DistributedDataParallel wrapper
        ckpt["state_dict"] = OrderedDict((k.replace("module.", ""), v) for k, v in ckpt["state_dict"].items())
    return ckpt


def load_params_from_ckpt(path, model, optimizer=None, logger=None):
    ckpt = get_ckpt(path)
    (model.module if hasattr(model, "module") else model).load_state_dict(ckpt["state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    if logger:
        logger("=> loaded checkpoint '%s' (epoch %s, commit %s)" % (path, ckpt["epoch"], ckpt.get("commit_id")))
    return model, optimizer, ckpt["epoch"]


def load_params_from_pretrain(path, model, strict=True, logger=None):
    ckpt = get_ckpt(path)
    (model.module if hasattr(model, "module") else model).load_state_dict(ckpt["state_dict"], strict=strict)
    if logger:
        logger("=> loaded pretrained model '%s' (epoch %s)" % (path, ckpt["epoch"]))
    return model


def load_metric_from_ckpt(path):
    ckpt = get_ckpt(path)
    return ckpt.get("metric"), ckpt["epoch"]


# ------------------------------------------------------------------------------------------------ meters
class DeviceMeters:
    """Loss and intersection / union / target histograms accumulated on the device (the reference syncs
    with the host three times per iteration for the same numbers)."""

    def __init__(self, n_classes, ignore_label, device):
        self.k, self.ignore = n_classes, ignore_label
        self.cnt = torch.zeros(3, n_classes, dtype=torch.int64, device=device)     # intersection, prediction area, target area
        self.loss = torch.zeros(2, dtype=torch.float64, device=device)   # sum of loss * n, sum of n

    @property
    def hist(self):
        """float64 [3, k]: intersection, union, target (the reference's three meters)."""
        c = self.cnt.to(torch.float64)
        return torch.stack((c[0], c[1] + c[2] - c[0], c[2]))

    @torch.no_grad()
    def update(self, loss, preds, labels, p2v=None):
        """No host synchronisation.  preds: class per point — or per VOXEL with p2v (the fused head's argmax: the gather rides in the
        kernel).  Device tensors: one launch (ops.seg_meters; three scatter_add_ over 800 k points onto 21 addresses took 612 us per
        iteration); otherwise fixed-shape torch ops (boolean-mask indexing and bincount would each read a size back).  Ignored /
        out-of-range labels are dropped, predictions clamped to [0, k - 1]."""
        k = self.k
        if labels.is_cuda and labels.dtype == torch.int64 and k <= 256 and preds.dtype in (torch.int32, torch.int64):
            from . import ops
            ops.seg_meters(self.cnt, preds, labels, k, self.ignore, p2v=p2v)
        else:
            if p2v is not None:
                preds = preds[p2v.long()]
            preds = preds.long()
            valid = (labels != self.ignore) & (labels >= 0) & (labels < k)
            t = torch.where(valid, labels, k)
            p = torch.where(valid, preds.clamp(0, k - 1), k)
            hit = torch.where(p == t, t, k)
            ones = torch.ones_like(t)
            cnt = torch.zeros(3, k + 1, dtype=torch.int64, device=t.device)
            cnt[0].scatter_add_(0, hit, ones)
            cnt[1].scatter_add_(0, p, ones)
            cnt[2].scatter_add_(0, t, ones)
            self.cnt += cnt[:, :k]
        n = float(labels.shape[0])
        self.loss[0] += loss.detach().double() * n
        self.loss[1] += n

    def all_reduce(self):
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.cnt)
            dist.all_reduce(self.loss)

    def read(self):
        """One host read-back: (mean loss, mIoU, mAcc, allAcc, per-class IoU)."""
        h = self.hist.cpu().numpy()
        l = self.loss.cpu().numpy()
        iou = h[0] / (h[1] + 1e-10)
        acc = h[0] / (h[2] + 1e-10)
        return float(l[0] / max(l[1], 1.0)), float(iou.mean()), float(acc.mean()), float(h[0].sum() / (h[2].sum() + 1e-10)), iou


# ------------------------------------------------------------------------------------------------ data
def make_loader(cfg, args, device, rank, world, epoch, split="train"):
    """Yields collate dictionaries resident on `device` (reference dataset/dataset.py:121-187 contract).
    Synthetic stand-in for the ScanNet / 3D-FRONT / S3DIS loaders: seeded procedural scenes, a different
    set per epoch, rank and split."""
    from .collate import collate_device
    from .scene import make_scene
    bs = args.batch_size
    n_batches = max(1, args.synthetic_scenes // (bs * world))
    base = {"train": 0, "target": 500000, "val": 900000}[split] + (0 if split == "val" else epoch * 10007)
    for b in range(n_batches):
        items = []
        for k in range(bs):
            sid = base + (b * world + rank) * bs + k
            xyz, xyz_mid, lab = make_scene(1000 + sid, args.synthetic_voxels, cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_scale)
            items.append((xyz, xyz_mid, lab, sid))
        yield collate_device(items, device, voxel_mode=cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_mode,
                             full_scale=cfg.DATA_CONFIG.DATA_PROCESSOR.get("full_scale", [128, 512]))


# ------------------------------------------------------------------------------------------------ training
class Trainer:
    def __init__(self, args, cfg, device, rank=0, world=1, log=print):
        from . import dist as ddist
        from .dsnorm import DSNorm
        from .model import PyramidPrefetcher, SparseConvNet
        from .spconv import functional as Fsp
        self.args, self.cfg, self.device, self.rank, self.world, self.log = args, cfg, device, rank, world, log
        model = SparseConvNet(cfg)
        if args.sync_bn:
            # tool/train.py:329-330, literally: torch's SyncBatchNorm (statistics all-gathered over the process
            # group, RCCL on GPUs).  It is not nn.BatchNorm1d, so SparseSequential applies it as a plain module:
            # torch's kernels, no conv-epilogue statistics.  As in the reference it takes precedence over DSNorm.
            model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        elif cfg.MODEL.get("dsnorm", False):
            model = DSNorm.convert_dsnorm(model)
        self.model = model.to(device)
        self.fdt = torch.float32 if args.dtype == "f32" else torch.bfloat16
        self.optimizer = build_optimizer(cfg.OPTIMIZATION, self.model)
        self.curve = []      # per epoch: mean training loss (+ the held-out metrics when evaluated): --curve_json
        self.deferred = Fsp.set_deferred_wgrad(True)
        self.reducer = None
        if world > 1:
            if self.deferred:
                self.reducer = ddist.GradAllReduce(self.model)
            else:
                self.model = ddist.wrap_ddp(self.model, device.index)
        self.with_pairs = bool(Fsp.WGRAD_PAIRS and self.fdt == torch.bfloat16)
        from .model import tile_levels_for
        self.with_tiles = tile_levels_for(self.fdt)
        net = self.model.module if hasattr(self.model, "module") else self.model
        self.n_levels = len(net.unet.nPlanes)
        self.prefetch = PyramidPrefetcher(device, self.n_levels) if device.type == "cuda" else None
        self.iters_done = 0
        self._loaders = {}
        self.step_times = []      # (iterations, seconds) of the steady part of every epoch (see train_epoch)

    # one forward + backward of one batch; `domain`: None | "source" | "target" (DSNorm statistics)
    def _pass(self, batch, pyramid, domain, weight=1.0):
        from .dsnorm import set_ds_source, set_ds_target
        from .model import point_predictions, voxelize_and_run
        if self.cfg.MODEL.get("dsnorm", False) and domain is not None:
            self.model.apply(set_ds_source if domain == "source" else set_ds_target)
        labels = batch["labels"]
        # head + CrossEntropyLoss at voxel level (no [points, classes] matrix); the meters' prediction = the voxel's argmax
        loss = voxelize_and_run(self.cfg, self.model, batch, self.device, feature_dtype=self.fdt, inputs_ready=True, pyramid=pyramid,
                                labels=labels, ignore_index=self.cfg.DATA_CONFIG.DATA_CLASS.ignore_label)
        (loss * weight if weight != 1.0 else loss).backward()
        # predictions for the meters: the fused head's per-VOXEL argmax with the point -> voxel map (the gather rides in the meters'
        # kernel), or a class per point
        net = self.model.module if hasattr(self.model, "module") else self.model
        vp, p2v = getattr(net, "voxel_pred", None), batch["p2v_map"]
        if vp is not None and p2v.is_cuda and p2v.dtype == torch.int32:
            return loss, (vp, p2v), labels
        return loss, (point_predictions(self.model, p2v), None), labels

    def _loader(self, split):
        """(iterable of host / device batches, object with set_epoch) per split, made once: the dataset resident in HBM
        (default) or a persistent DataLoader with worker processes (--host_loader; reference dataset/__init__.py:62-75)."""
        from . import dist as ddist
        from .loader import DeviceScenes, host_loader, synthetic_dataset
        if split not in self._loaders:
            if self.rank == 0:
                ds = synthetic_dataset(self.cfg, self.args, split)      # (generates missing base scenes)
            ddist.barrier()
            if self.rank != 0:
                ds = synthetic_dataset(self.cfg, self.args, split)
            seed = self.args.manual_seed or 0
            fs0 = self.cfg.DATA_CONFIG.DATA_PROCESSOR.get("full_scale", [128, 512])[0]
            if self.args.host_loader:   # (same sampler seed and the same spatial-shape clip as the HBM-resident loader below)
                self._loaders[split] = host_loader(ds, self.args.batch_size, self.rank, self.world, self.args.workers,
                                                   shuffle=split != "val", seed=ds.seed + seed, full_scale0=fs0)
            else:
                dsc = DeviceScenes(ds.paths, ds.length, ds.voxel_scale, ds.seed + seed, self.args.batch_size, self.rank, self.world,
                                   self.device, augment=ds.augment, shuffle=split != "val",
                                   full_scale0=fs0)
                self._loaders[split] = (dsc, dsc)
        return self._loaders[split]

    def _batches(self, epoch, split):
        """(batch, prebuilt rulebooks): worker processes produce the scenes, a feeder thread collates them on the device and
        builds their rulebooks two batches ahead (doda_amd.loader); --inline_loader: everything in this thread, as before."""
        from .model import PyramidPrefetcher
        if not self.args.inline_loader and self.device.type == "cuda":
            from .loader import DeviceFeeder
            dl, sampler = self._loader(split)
            sampler.set_epoch(epoch)
            feeder = DeviceFeeder(iter(dl), self.device, prefetcher=self.prefetch, with_pairs=self.with_pairs,
                                  with_tiles=self.with_tiles, voxel_mode=self.cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_mode,
                                  full_scale=self.cfg.DATA_CONFIG.DATA_PROCESSOR.get("full_scale", [128, 512]),
                                  voxel_order=getattr(self, "voxel_order", None) or getattr(self.args, "voxel_order", "auto"))
            try:
                for pair in feeder:
                    yield pair
            finally:
                feeder.close()
                self.voxel_order = feeder.voxel_order      # ("auto" is decided once, from the first batch)
                if feeder.batches:
                    self.feeder_ms = (1e3 * feeder.wait_host_s / feeder.batches, 1e3 * feeder.collate_s / feeder.batches)
            return
        it = make_loader(self.cfg, self.args, self.device, self.rank, self.world, epoch, split)
        nxt = next(it, None)
        fut = self.prefetch.submit(nxt, self.with_pairs, self.with_tiles, now=True) if (nxt is not None and self.prefetch) else None
        while nxt is not None:
            cur, cur_fut = nxt, fut
            nxt = next(it, None)
            fut = self.prefetch.submit(nxt, self.with_pairs, self.with_tiles) if (nxt is not None and self.prefetch) else None
            yield cur, (PyramidPrefetcher.take(cur_fut, self.device) if cur_fut is not None else None)

    def train_epoch(self, epoch, total_epochs):
        args, cfg = self.args, self.cfg
        self.model.train()
        meters = DeviceMeters(cfg.COMMON_CLASSES.n_classes, cfg.DATA_CONFIG.DATA_CLASS.ignore_label, self.device)
        n_iter = max(1, args.synthetic_scenes // (args.batch_size * self.world))
        target = self._batches(epoch, "target") if args.self_train else None
        t0 = time.time()
        warm = min(getattr(args, "timing_warmup", 10), max(0, n_iter - 2))
        t_steady = None
        for i, (batch, pyramid) in enumerate(self._batches(epoch, "train")):
            if i == warm:   # steady-state clock: from here to the end of the epoch, device drained on both sides
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                t_steady = (time.time(), i)
                vox_steady = 0
            if t_steady is not None:
                vox_steady += int(batch["voxel_locs"].shape[0])
            lr = adjust_lr(cfg.OPTIMIZATION, self.optimizer, None, total_epochs, n_iter, epoch, i)
            self.optimizer.zero_grad(set_to_none=True)
            if args.self_train:   # tool/st.py:136-198: source pass, then target pass, ONE optimizer step
                st = cfg.get("SELF_TRAIN", Config())
                loss, preds, labels = self._pass(batch, pyramid, "source",
                                                 st.get("SRC", Config()).get("loss_weight", 1.0))
                tb, tp = next(target)
                if self.reducer is not None:
                    self.reducer.arm()      # (the second backward pass completes the gradients: its hook may start the exchange)
                self._pass(tb, tp, "target", st.get("TAR", Config()).get("loss_weight", 1.0))
            else:
                if self.reducer is not None:
                    self.reducer.arm()
                loss, preds, labels = self._pass(batch, pyramid, "source")
            if self.reducer is not None:
                self.reducer.reduce()
            if cfg.OPTIMIZATION.get("clip_grad", False):
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=10)
            self.optimizer.step()
            meters.update(loss, preds[0], labels, p2v=preds[1])
            self.iters_done += 1
            if (i + 1) % args.print_freq == 0 or i == n_iter - 1:
                l, miou, macc, allacc, _ = meters.read()   # the iteration's only host read-back
                self.log("Epoch: [%d/%d][%d/%d] Batch %.3f s Loss %.4f Accuracy %.4f lr %.6f" % (
                    epoch + 1, total_epochs, i + 1, n_iter, (time.time() - t0) / (i + 1), l, allacc, lr))
            if args.max_iters is not None and self.iters_done >= args.max_iters:
                break
        if t_steady is not None:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            done = i + 1 - t_steady[1]
            if done > 0:
                self.step_times.append((done, time.time() - t_steady[0], vox_steady))
                self.log("Steady state: %d iterations, %.3f ms per iteration" % (done, 1e3 * self.step_times[-1][1] / done))
        meters.all_reduce()
        l, miou, macc, allacc, _ = meters.read()
        self.log("Train result at epoch [%d/%d]: mIoU/mAcc/allAcc %.4f/%.4f/%.4f." % (epoch + 1, total_epochs, miou, macc, allacc))
        self.curve.append({"epoch": epoch + 1, "iterations": self.iters_done, "train_loss": l, "train_miou": miou})
        return l

    @torch.no_grad()
    def validate_epoch(self, epoch):
        from .dsnorm import set_ds_target
        from .model import cross_entropy, voxelize_and_run
        cfg = self.cfg
        self.model.eval()
        if self.reducer is not None:
            self.reducer.sync_buffers()   # evaluation with rank 0's running statistics on every rank
        if cfg.MODEL.get("dsnorm", False):
            self.model.apply(set_ds_target)
        meters = DeviceMeters(cfg.COMMON_CLASSES.n_classes, cfg.DATA_CONFIG.DATA_CLASS.ignore_label, self.device)
        for batch, pyramid in self._batches(epoch, "val"):
            scores = voxelize_and_run(cfg, self.model, batch, self.device, feature_dtype=self.fdt,
                                      inputs_ready=True, pyramid=pyramid)
            loss = cross_entropy(scores, batch["labels"], ignore_index=cfg.DATA_CONFIG.DATA_CLASS.ignore_label)
            meters.update(loss, scores.argmax(1), batch["labels"])
        meters.all_reduce()
        l, miou, macc, allacc, iou = meters.read()
        self.log("Val result: mIoU/mAcc/allAcc %.4f/%.4f/%.4f." % (miou, macc, allacc))
        self.log("Val IoU per class: " + " ".join("%.4f" % v for v in iou))
        if self.curve and self.curve[-1]["epoch"] == epoch + 1:
            self.curve[-1].update({"val_loss": l, "val_miou": miou, "val_allacc": allacc, "val_iou": [float(v) for v in iou]})
        return miou


def main(argv=None):
    from . import dist as ddist
    args, cfg = parse_config(argv)
    world, rank, local_rank = (1, 0, 0)
    if args.launcher != "none":
        world, rank, local_rank = ddist.setup()
        cfg.LOCAL_RANK = local_rank
    if not torch.cuda.is_available():
        raise RuntimeError("doda_amd.train needs an MI355X: the native ops have no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    from .host import pin_to_device_numa
    pin_to_device_numa(device.index)     # issuing threads on the GPU's NUMA node (doda_amd/host.py)
    if args.batch_size is None:
        args.batch_size = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU
    else:
        assert args.batch_size % world == 0, "Batch size should match the number of gpus"
        args.batch_size //= world
    args.epochs = cfg.OPTIMIZATION.NUM_EPOCHS if args.epochs is None else args.epochs
    if args.manual_seed is not None:
        np.random.seed(args.manual_seed)
        torch.manual_seed(args.manual_seed)
        torch.cuda.manual_seed_all(args.manual_seed)
    root = Path(args.output_root) if args.output_root else Path(os.getcwd()) / "output"
    output_dir = root / cfg.EXP_GROUP_PATH / cfg.TAG / args.extra_tag
    ckpt_dir = output_dir / "ckpt"
    if rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    ddist.barrier()

    def log(msg):
        if rank == 0:
            print(msg, flush=True)

    trainer = Trainer(args, cfg, device, rank, world, log)
    log("#classifier parameters: %d" % sum(p.nelement() for p in trainer.model.parameters()))
    best_miou, best_epoch = 0.0, 0
    if args.weight:
        load_params_from_pretrain(args.weight, trainer.model, strict=not args.pretrain_not_strict, logger=log)
    if args.resume:
        _, _, args.start_epoch = load_params_from_ckpt(args.resume, trainer.model, trainer.optimizer, log)
    else:
        ckpts = sorted(glob.glob(str(ckpt_dir / "*train_epoch_*.pth")), key=os.path.getmtime)
        if ckpts:
            _, _, args.start_epoch = load_params_from_ckpt(ckpts[-1], trainer.model, trainer.optimizer, log)
    if (ckpt_dir / "best_train.pth").exists():
        best_miou, best_epoch = load_metric_from_ckpt(str(ckpt_dir / "best_train.pth"))
        best_miou = best_miou or 0.0
    log("optimizer LR: %s" % trainer.optimizer.param_groups[0]["lr"])
    for epoch in range(args.start_epoch, args.epochs):
        trainer.train_epoch(epoch, args.epochs)
        epoch_log = epoch + 1
        if rank == 0 and epoch_log % args.ckpt_save_freq == 0:
            filename = ckpt_dir / ("train_epoch_%d.pth" % epoch_log)
            log("Saving checkpoint to: %s" % filename)
            save_params(filename, trainer.model, trainer.optimizer, epoch_log)
            if not args.reserve_old_ckpt:
                old = ckpt_dir / ("train_epoch_%d.pth" % (epoch_log - args.ckpt_save_freq * 2))
                if old.exists():
                    old.unlink()
        if cfg.get("EVALUATION", Config()).get("evaluate", False) and epoch_log % cfg.EVALUATION.eval_freq == 0:
            miou = trainer.validate_epoch(epoch)
            if rank == 0 and miou > best_miou:
                best_miou, best_epoch = miou, epoch_log
                save_params(ckpt_dir / "best_train.pth", trainer.model, trainer.optimizer, epoch_log, metric=best_miou)
        log("Best epoch: %d, best mIoU: %s" % (best_epoch, best_miou))
        if args.max_iters is not None and trainer.iters_done >= args.max_iters:
            break
    if trainer.prefetch is not None:
        trainer.prefetch.shutdown()
    if trainer.reducer is not None:
        trainer.reducer.close()      # the gradient homes live in a process-wide map: drop this reducer's
    if args.timing_json and rank == 0 and trainer.step_times:
        import json
        its = sum(t[0] for t in trainer.step_times)
        sec = sum(t[1] for t in trainer.step_times)
        vox = sum(t[2] for t in trainer.step_times)
        with open(args.timing_json, "w") as f:
            json.dump({"iterations": its, "seconds": sec, "ms_per_iter": 1e3 * sec / its, "world": world,
                       "voxels_per_s": vox * world / sec if sec > 0 else None,   # (this rank's voxels x ranks: weak scaling)
                       "batch_size_per_gpu": args.batch_size, "workers": args.workers, "dtype": args.dtype,
                       "voxels_per_scene": args.synthetic_voxels,
                       "feeder_wait_for_workers_ms": getattr(trainer, "feeder_ms", (None, None))[0],
                       "feeder_collate_and_rulebooks_ms": getattr(trainer, "feeder_ms", (None, None))[1]}, f)
    if args.curve_json and rank == 0:
        import json
        with open(args.curve_json, "w") as f:
            json.dump({"dtype": args.dtype, "seed": args.manual_seed, "batch_size_per_gpu": args.batch_size, "world": world,
                       "scenes_per_epoch": args.synthetic_scenes, "voxels_per_scene": args.synthetic_voxels, "curve": trainer.curve}, f)
    ddist.barrier()


if __name__ == "__main__":
    main()
