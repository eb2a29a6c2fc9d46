#!/usr/bin/env python
"""Is doda_tilebook_build deterministic on a Z-ordered 1 cm scene (tiles whose rows span more than the bitmap form covers take
the hash + bitonic-sort path)?  Builds the level-1 / level-2 tilebooks N times and compares the bytes; counts the tiles per path.
usage: tbrace.py [reps=300] [scenes=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from doda_amd import ops, spconv
from doda_amd.collate import reorder_voxels
from doda_amd.scene import make_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, 500000, 1000, 100), os.environ.get("ORDER", "morton"))
idx = b["voxel_locs"].int().to(dev)
shape = [int(s) for s in b["spatial_shape"]]
t = spconv.SparseConvTensor(None, idx, shape, ns)
books = spconv.ops.build_pyramid(t, 3, with_pairs=False, with_tiles=0)
for key in ("subm1", "subm2"):
    tbl = books[key].tbl.contiguous()
    m = tbl.shape[1]
    tc = tbl.cpu().numpy()
    nt = (m + 255) // 256
    span = np.zeros(nt, dtype=np.int64)
    for k in range(nt):
        v = tc[:, k * 256:(k + 1) * 256]
        v = v[v >= 0]
        span[k] = int(v.max()) - int(v.min()) + 1 if v.size else 0
    hashed = int((span > 196608).sum())
    first = ops.tilebook_build(tbl)
    torch.cuda.synchronize()
    ref = first.clone()
    bad = 0
    for r in range(reps):
        tb = ops.tilebook_build(tbl)
        if not torch.equal(tb, ref):
            bad += 1
            if bad <= 3:
                diff = (tb != ref).nonzero().flatten()
                print("   rep %d: %d bytes differ, first at %d" % (r, diff.numel(), int(diff[0])), flush=True)
    print("%s: %d rows, %d tiles, %d on the hash path (span > 196608 rows); %d of %d rebuilds differ from the first" % (key, m, nt, hashed, bad, reps), flush=True)
