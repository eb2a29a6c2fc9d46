#!/usr/bin/env python
"""conv_wlds48 (48 -> 48, level-3 size of the bench batch) in the step's instantiation (statistics + residual; data gradient with
the BatchNorm-backward statistics), cold over rotating buffer sets: us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
t = spconv.SparseConvTensor(None, idx, shape, 4)
books = spconv.ops.build_pyramid(t, 4, with_pairs=False, with_tiles=2)
sub = books["subm3"]
m = sub.tbl.shape[1]
c = 48
w = torch.randn(27, c, c, device=dev) * 0.05
plan = ops.PackPlan([(w, 27, c, c, 0, 2), (w, 27, c, c, 2, 2)], dev)
plan.run()
sets = [(torch.randn(m, c, device=dev).bfloat16(), torch.randn(m, c, device=dev).bfloat16(), torch.randn(m, c, device=dev).bfloat16(),
         sub.tbl.clone()) for _ in range(24)]
mean = torch.zeros(c, device=dev); invstd = torch.ones(c, device=dev); gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
def run(kind, k):
    x, gy, res, tbl = sets[k % len(sets)]
    if kind == "fwd":
        ops.spconv_gather(x, None, tbl, m, 0, c, packed=plan.outputs[0], residual=res, want_stats=True)
    else:
        ops.spconv_gather(gy, None, tbl, m, 2, c, packed=plan.outputs[1], want_stats=True, bn=(x, mean, invstd, gamma, beta, True))
for kind in ("fwd", "dgrad"):
    for k in range(10): run(kind, k)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for k in range(200): run(kind, k)
    ev[1].record(); torch.cuda.synchronize()
    print("level 3 (%d rows) 48 -> 48 %s, step form, cold: %.1f us per launch" % (m, kind, ev[0].elapsed_time(ev[1]) * 1e3 / 200), flush=True)
