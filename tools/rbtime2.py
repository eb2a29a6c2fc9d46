#!/usr/bin/env python
"""Rulebook pyramid build time (13 rulebooks of the bench batch, tilebooks + strided pair lists) on an idle GPU.
usage: [DODA_RULEBOOK_GRID=0] python tools/rbtime2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
def build():
    t = spconv.SparseConvTensor(None, idx, shape, 4)
    spconv.ops.build_pyramid(t, 7, with_pairs=True, with_tiles=2)
for _ in range(3): build()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): build()
torch.cuda.synchronize()
print("pyramid build: %.3f ms (DODA_RULEBOOK_GRID=%s)" % ((time.perf_counter() - t0) / 20 * 1e3, os.environ.get("DODA_RULEBOOK_GRID", "1")))
