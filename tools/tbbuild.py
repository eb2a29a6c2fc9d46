"""doda_tilebook_build alone on the bench batch's level-1 and level-2 SubM tables (HIP events, 30 builds each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
for lvl in (1, 2):
    sub = spconv.ops.build_subm(idx, 4, shape, 3)
    m = idx.shape[0]
    for _ in range(3): ops.tilebook_build(sub.tbl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.tilebook_build(sub.tbl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("level %d: %d rows, tilebook_build (+ its 8-byte memset) %.1f us = %.2f TB/s of 108 M + 73 M bytes" % (lvl, m, us, 181.0 * m / us / 1e6), flush=True)
    down = spconv.ops.build_down2(idx, 4, shape, 2, 2, 0, 1)
    idx, shape = down.outids, down.out_spatial_shape
