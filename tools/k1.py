#!/usr/bin/env python
"""One kernel under the profiler: SubM 16->16 gather forward (level-1 size), N launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
if os.environ.get("K1_LIB"):      # a variant build of the library (experiments)
    from doda_amd import _lib
    _lib.LIB_PATH = os.environ["K1_LIB"]
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
c = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
cache = "/tmp/k1_batch.pt"
if os.path.exists(cache):
    batch = torch.load(cache, weights_only=False)
else:
    batch = make_batch(4, 150000, 1000)
    torch.save(batch, cache)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
sub = spconv.ops.build_subm(idx, 4, shape, 3)
m = idx.shape[0]
tdt = torch.float32 if dt == "f32" else torch.bfloat16
x = torch.randn(m, c, device=dev).to(tdt); gy = torch.randn(m, c, device=dev).to(tdt)
w = torch.randn(27, c, c, device=dev) * 0.05
s = 4 if dt == "f32" else 2
plan = ops.PackPlan([(w, 27, c, c, 0, s), (w, 27, c, c, 2, s)], dev); plan.run()
pairs = sub.wgrad_lists() if which == "wgradp" else None
res = torch.randn(m, c, device=dev).to(tdt)
tb = ops.tilebook_build(sub.tbl) if (c == 16 and os.environ.get("DODA_NO_TILE", "0") != "1") else None
for _ in range(reps):
    if which == "fwd": ops.spconv_gather(x, None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb)
    elif which == "dgrad": ops.spconv_gather(gy, None, sub.tbl, m, 2, c, packed=plan.outputs[1], tilebook=tb)
    elif which == "fwdstep":   # the instantiation the training step launches: statistics + residual in the epilogue
        ops.spconv_gather(x, None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=res, want_stats=True)
    elif which == "fwdtot":    # ... with the statistics as fp64 totals (ABI 9: what the step launches by default)
        if "big_tot" not in globals():
            big_tot = torch.zeros(4 * 8 * 2 * c, dtype=torch.float64, device=dev)      # (room for a padded-layout experiment)
        ops.spconv_gather(x, None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=res,
                          want_stats=big_tot[:8 * 2 * c].view(8, 2, c))
    elif which == "wgradt": ops.spconv_wgrad_multi([(x, gy, sub.tbl, m, None, None, tb)] * 8)   # LDS-staged tile kernel, 8 layers per call
    elif which == "wgradp": ops.spconv_wgrad_multi([(x, gy, sub.tbl, m, pairs)] * 8)   # pair-list kernel, 8 layers per call
    else: ops.spconv_wgrad_multi([(x, gy, sub.tbl, m)] * 8)                              # gather-table kernel
torch.cuda.synchronize()
print("done", m)
