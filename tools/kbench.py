#!/usr/bin/env python
"""Kernel micro-bench on the U-Net's own level geometry: for every level of a synthetic batch,
time SubM gather fwd / dgrad / wgrad (c -> c) and the level's down2 conv, per dtype.
  python tools/kbench.py [--scenes 4] [--voxels 150000] [--dtypes bf16,f32] [--levels 1,2,3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from doda_amd import ops, spconv  # noqa: E402
from doda_amd.scene import make_batch  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--voxels", type=int, default=150000)
    ap.add_argument("--dtypes", default="bf16,f32")
    ap.add_argument("--levels", default="1,2,3,4,5,6,7")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--order", default="scene", help="voxel order: scene | morton | random | zyx")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    batch = make_batch(a.scenes, a.voxels, 1000)
    idx = batch["voxel_locs"].int().to(dev)
    shape = [int(s) for s in batch["spatial_shape"]]
    if a.order != "scene":
        b, x, y, z = [idx[:, k].long() for k in range(4)]
        if a.order == "random":
            perm = torch.randperm(idx.shape[0], device=dev)
        elif a.order == "zyx":
            perm = torch.argsort(((b * 4096 + x) * 4096 + y) * 4096 + z)
        else:
            def spread(v):
                r = torch.zeros_like(v)
                for bit in range(12):
                    r |= ((v >> bit) & 1) << (3 * bit)
                return r
            perm = torch.argsort((b << 40) | (spread(x) << 2) | (spread(y) << 1) | spread(z))
        idx = idx[perm].contiguous()
    levels = [int(v) for v in a.levels.split(",")]
    for lvl in range(1, 8):
        m = idx.shape[0]
        c = 16 * lvl
        sub = spconv.ops.build_subm(idx, a.scenes, shape, 3)
        down = spconv.ops.build_down2(idx, a.scenes, shape, 2, 2, 0, 1) if lvl < 7 else None
        if lvl in levels:
            pairs = int((sub.tbl >= 0).sum())
            for dt in a.dtypes.split(","):
                tdt = torch.float32 if dt == "f32" else torch.bfloat16
                s = 4 if dt == "f32" else 2
                x = torch.randn(m, c, device=dev).to(tdt)
                gy = torch.randn(m, c, device=dev).to(tdt)
                w = torch.randn(27, c, c, device=dev) * 0.05
                tf = timed(lambda: ops.spconv_gather(x, w, sub.tbl, m, 0, c), a.reps)
                td = timed(lambda: ops.spconv_gather(gy, w, sub.tbl, m, 2, c), a.reps)
                tw = timed(lambda: ops.spconv_wgrad(x, gy, sub.tbl, m), a.reps)
                bf = s * 2 * m * c + 4 * 27 * c * c + 8 * pairs
                line = "L%d M=%7d c=%3d %-4s  fwd %8.1f us (%5.0f GB/s)  dgrad %8.1f  wgrad %8.1f" % (
                    lvl, m, c, dt, tf, bf / tf / 1e3, td, tw)
                if dt == "f32" and c == 16:   # the tile kernel's fp32 mode over the rulebook's tilebook
                    tb = ops.tilebook_build(sub.tbl)
                    tt = timed(lambda: ops.spconv_gather(x, w, sub.tbl, m, 0, c, tilebook=tb), a.reps)
                    line += "  fwd over the tilebook %7.1f" % tt
                if dt == "bf16" and c % 16 == 0:
                    te = timed(lambda: ops.rulebook_pairs(sub.tbl, m, True, pad=False), a.reps)
                    pr, num, seg = ops.rulebook_pairs(sub.tbl, m, True, pad=False, with_seg=True)
                    tp = timed(lambda: ops.spconv_wgrad_pairs(x, gy, pr[0], pr[1], num, seg), a.reps)
                    line += " wgrad-pairs %7.1f (%5.0f GB/s; list export %6.1f)" % (tp, bf / tp / 1e3, te)
                if down is not None:
                    mo = down.outids.shape[0]
                    wd = torch.randn(8, c, c + 16, device=dev) * 0.05
                    gyo = torch.randn(mo, c + 16, device=dev).to(tdt)
                    t1 = timed(lambda: ops.spconv_gather(x, wd, down.tbl, mo, 0, c + 16), a.reps)
                    t2 = timed(lambda: ops.spconv_gather(gyo, wd, down.tbl_rev, m, 1, c), a.reps)
                    t3 = timed(lambda: ops.spconv_wgrad(x, gyo, down.tbl, mo), a.reps)
                    line += "  | down fwd %7.1f dgrad(=inverse fwd) %7.1f wgrad %7.1f" % (t1, t2, t3)
                print(line, flush=True)
        if down is None:
            break
        idx, shape = down.outids, down.out_spatial_shape
    tr = timed(lambda: spconv.ops.build_subm(batch["voxel_locs"].int().to(dev), a.scenes,
                                             [int(s) for s in batch["spatial_shape"]], 3), 10)
    print("rulebook subm L1 build: %.1f us" % tr)


if __name__ == "__main__":
    main()
