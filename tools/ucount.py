#!/usr/bin/env python
"""Distinct-neighbour-row counts per 256-row tile (tilebook ucount) of a synthetic scene: what share of the tiles exceeds the
tile kernels' staging capacities.  usage: ucount.py [target_voxels=150000] [voxel_scale=50] [scenes=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from doda_amd import spconv
from doda_amd.scene import make_batch
tv = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
vs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
b = make_batch(ns, tv, 1000, voxel_scale=vs, full_scale=(128, 2048))
idx = b["voxel_locs"].int().to(dev)
shape = [int(s) for s in b["spatial_shape"]]
t = spconv.SparseConvTensor(None, idx, shape, ns)
books = spconv.ops.build_pyramid(t, 3, with_pairs=False, with_tiles=0)
for key in ("subm1", "subm2"):
    tbl = books[key].tbl.clone()
    present = (tbl >= 0).sum().item() / tbl.shape[1]
    # distinct rows per tile straight from the table (the builder keeps no count above its list capacity)
    m = tbl.shape[1]; nt = (m + 255) // 256
    cnt = np.zeros(nt, dtype=np.int64)
    tc = tbl.cpu().numpy()
    for k in range(nt):
        v = tc[:, k * 256:(k + 1) * 256].ravel()
        cnt[k] = np.unique(v[v >= 0]).size
    q = np.percentile(cnt, [50, 90, 99, 100])
    print("%s: %d rows, %d tiles, %.1f pairs/row; distinct rows per tile p50 %d p90 %d p99 %d max %d; tiles > 640: %.2f %%, > 704: %.2f %%, > 960: %.2f %%, > 1023: %.2f %%, > 1216: %.2f %%" % (
        key, m, nt, present, q[0], q[1], q[2], q[3], 100.0 * (cnt > 640).mean(), 100.0 * (cnt > 704).mean(), 100.0 * (cnt > 960).mean(), 100.0 * (cnt > 1023).mean(), 100.0 * (cnt > 1216).mean()), flush=True)
