#!/usr/bin/env python
"""Voxel numbering (first-appearance order of the reference against the Z-order renumbering of doda_amd.collate.reorder_voxels):
distinct rows per 256-row tile at levels 1-2, and the bench step on both.  usage: orderab.py [voxels=150000] [voxel_scale=50] [scenes=4] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
from doda_amd.spconv import functional as Fsp
tv = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
vs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda:0")
base = make_batch(ns, tv, 1000, voxel_scale=vs)
cfg = default_cfg()
Fsp.set_deferred_wgrad(True)
for order in ("first", "morton", "first", "morton"):
    b = reorder_voxels(base, order)
    spconv.ops._tile_state["skip"] = 0          # (the overflow back-off of the previous arm)
    idx = b["voxel_locs"].int().to(dev)
    shape = [int(s) for s in b["spatial_shape"]]
    t = spconv.SparseConvTensor(None, idx, shape, ns)
    books = spconv.ops.build_pyramid(t, 3, with_pairs=False, with_tiles=0)
    line = []
    for key in ("subm1", "subm2"):
        tc = books[key].tbl.cpu().numpy()
        m = tc.shape[1]; nt = (m + 255) // 256
        cnt = np.array([np.unique(tc[:, k * 256:(k + 1) * 256][tc[:, k * 256:(k + 1) * 256] >= 0]).size for k in range(0, nt, max(1, nt // 400))])
        line.append("%s %d rows: distinct/tile p50 %d p90 %d max %d, > 1023: %.1f %%" % (key, m, *np.percentile(cnt, [50, 90, 100]), 100.0 * (cnt > 1023).mean()))
    torch.manual_seed(0)
    net = SparseConvNet(cfg).to(dev).train()
    opt = FusedSGD(net.parameters(), lr=1e-3, momentum=0.9) if "FusedSGD" in globals() else torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    def step():
        opt.zero_grad(set_to_none=True)
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True), bd["labels"], ignore_index=255)
        loss.backward()
        opt.step()
        return loss
    for _ in range(10):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    print("%-6s %.3f ms/step (rulebooks built in line)  loss %.5f | %s" % (order, (time.perf_counter() - t0) / steps * 1e3, float(loss), " | ".join(line)), flush=True)
