#!/usr/bin/env python
"""Roofline figures of the kernels beside the U-Net: device voxelize_idx, voxel pooling fwd/bwd,
knnquery (k = 1, the only call DODA makes: model/unet.py:136) and ballquery, on the bench batch
(4 scenes x ~200 k points).  HIP-event times, algorithmic bytes / distance evaluations as stated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import ops
from doda_amd.scene import make_batch

dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
locs = batch["locs"].to(dev)                       # int64 [N, 4] (batch, x, y, z)
n = locs.shape[0]


def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


oc, imap, omap = ops.voxelize_idx_device(locs, 4, 4)
m, ma = omap.shape[0], omap.shape[1] - 1
us = timed(lambda: ops.voxelize_idx_device(locs, 4, 4))
alg = n * (32 + 4) + m * (32 + 4 * (ma + 1))       # coords in, point map out, voxel coords + voxel->point map out
print("voxelize_idx (device, incl. its size read-back): N %d -> M %d, max_active %d: %.0f us, %.1f GB/s algorithmic (%.3f of 8 TB/s)"
      % (n, m, ma, us, alg / us / 1e3, alg / us / 1e3 / 8000))
for c in (3, 6):
    feats = torch.randn(n, c, device=dev); out = torch.zeros(m, c, device=dev)
    us = timed(lambda: ops.voxelize_fp(feats, out, omap, 4, m, ma, c))
    alg = 4 * (ma + 1) * m + 4 * c * (n + m)
    print("voxelize_fp  C=%d: %.1f us, %.0f GB/s (%.3f)" % (c, us, alg / us / 1e3, alg / us / 1e3 / 8000))
    dout = torch.randn(m, c, device=dev); dfe = torch.zeros(n, c, device=dev)
    us = timed(lambda: ops.voxelize_bp(dout, dfe, omap, 4, m, ma, c))
    print("voxelize_bp  C=%d: %.1f us, %.0f GB/s (%.3f)" % (c, us, alg / us / 1e3, alg / us / 1e3 / 8000))

# knnquery k = 1: every point of the full cloud looks up its nearest point of the cropped cloud (same scene)
xyz_all = batch["locs_float"].to(dev).contiguous()
offs_all = batch["offsets"].to(dev).int()
keep = torch.rand(n, device=dev) < 0.5
bidx = locs[:, 0]
xyz = xyz_all[keep].contiguous()
offs = torch.zeros(5, dtype=torch.int32, device=dev)
offs[1:] = torch.cumsum(torch.bincount(bidx[keep], minlength=4), 0).int()
idx = torch.zeros(n, 1, dtype=torch.int32, device=dev); d2 = torch.zeros(n, 1, device=dev)
us = timed(lambda: ops.knnquery(n, 1, xyz, xyz_all, offs[1:].contiguous(), offs_all[1:].contiguous(), idx, d2), reps=3)
per_scene_q = torch.bincount(bidx, minlength=4).double(); per_scene_c = torch.bincount(bidx[keep], minlength=4).double()
evals = float((per_scene_q * per_scene_c).sum())
# 3 sub + 3 mul + 2 add + compare + 2 selects per evaluation; 256 CUs x 4 SIMD x 16 lanes x 2.4 GHz lane-ops/s
print("knnquery k=1: %d queries x ~%d candidates: %.2f ms, %.1f G distance evaluations/s = %.2f of the fp32 VALU issue rate at 11 ops each"
      % (n, int(per_scene_c.mean()), us / 1e3, evals / us / 1e3, evals * 11 / (us * 1e-6) / (256 * 4 * 16 * 2.4e9)))
sub = torch.arange(0, n, 8, device=dev)
pts = xyz_all[sub].contiguous(); b8 = bidx[sub].int().contiguous()
boff = torch.zeros(5, dtype=torch.int32, device=dev); boff[1:] = torch.cumsum(torch.bincount(b8.long(), minlength=4), 0).int()
n8 = pts.shape[0]
bi = torch.zeros(n8 * 50, dtype=torch.int32, device=dev); sl = torch.zeros(n8, 2, dtype=torch.int32, device=dev)
us = timed(lambda: ops.ballquery_batch_p(pts, b8, boff, bi, sl, n8, 50, 0.03), reps=3)
ev = float((torch.bincount(b8.long(), minlength=4).double() ** 2).sum()) * 2   # count pass + fill pass
print("ballquery r=0.03, meanActive 50: %d points: %.2f ms, %.1f G distance evaluations/s" % (n8, us / 1e3, ev / us / 1e3))
