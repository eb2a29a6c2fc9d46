"""conv_tile MODE 1 (32 -> 32) at the level-2 size of the bench batch, statistics + residual epilogue, cold and warm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
down = spconv.ops.build_down2(idx, 4, shape, 2, 2, 0, 1)
idx2, shape2 = down.outids, down.out_spatial_shape
sub = spconv.ops.build_subm(idx2, 4, shape2, 3)
m = idx2.shape[0]; c = 32
tb = ops.tilebook_build(sub.tbl)
NSET = 8
xs = [torch.randn(m, c, device=dev).bfloat16() for _ in range(NSET)]
ys = [torch.empty_like(x) for x in xs]; rs = [torch.randn(m, c, device=dev).bfloat16() for _ in range(NSET)]
w = torch.randn(27, c, c, device=dev) * 0.05
plan = ops.PackPlan([(w, 27, c, c, 0, 2)], dev); plan.run()
def timed(fn, n=60):
    for k in range(5): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cold in (0, 1):
    sel = (lambda k: k % NSET) if cold else (lambda k: 0)
    t = timed(lambda k: ops.spconv_gather(xs[sel(k)], None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=rs[sel(k)], want_stats=True, out=ys[sel(k)]))
    d = timed(lambda k: ops.spconv_gather(xs[sel(k)], None, sub.tbl, m, 0, c, packed=plan.outputs[0], residual=rs[sel(k)], want_stats=True, out=ys[sel(k)]))
    print("L2 m=%d tiles=%d %s: conv_tile %.1f us, dense-table kernel %.1f us" % (m, (m + 255) // 256, "cold" if cold else "warm", t, d), flush=True)
