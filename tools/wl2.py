#!/usr/bin/env python
"""Weight gradient of the level-1 / level-2 SubM layers of the bench batch: tile blocks (wgrad_dma16) against the pair-list kernel,
us per layer, 8 layers per call.  usage: wl2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
books = {}
for ns in (4, 1):
    batch = make_batch(ns, 150000, 1000)
    idx = batch["voxel_locs"].int().to(dev)
    shape = [int(s) for s in batch["spatial_shape"]]
    t = spconv.SparseConvTensor(None, idx, shape, ns)
    books[ns] = spconv.ops.build_pyramid(t, 3, with_pairs=True, with_tiles=2)
for ns, key, c in ((4, "subm1", 16), (4, "subm1", 32), (4, "subm2", 32), (1, "subm1", 16), (1, "subm2", 32)):
    data = books[ns][key]
    m = data.tbl.shape[1]
    from doda_amd._ext import ext
    tb_ok = ext.has_tilebook(data.tbl)
    tbl_plain = data.tbl.clone()
    tb = ops.tilebook_build(tbl_plain)
    pairs = data.wgrad_lists()
    x = torch.randn(m, c, device=dev).bfloat16(); gy = torch.randn(m, c if key == "subm2" else 16, device=dev).bfloat16()
    for name, job in (("tile ", (x, gy, tbl_plain, m, None, None, tb)), ("pairs", (x, gy, tbl_plain, m, pairs, None, None))):
        jobs = [job] * 8
        for _ in range(3): ops.spconv_wgrad_multi(jobs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.spconv_wgrad_multi(jobs)
        torch.cuda.synchronize()
        print("%s rows %7d  %2d -> %2d  %s  %.1f us per layer" % (key, m, c, gy.shape[1], name, (time.perf_counter() - t0) / 10 / 8 * 1e6), flush=True)
