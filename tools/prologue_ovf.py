import sys; sys.path.insert(0, "/root/repo")
import torch
from doda_amd import ops, spconv
from doda_amd._ext import ext
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
def timed(fn, n=30):
    for k in range(3): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for seed, scale in ((1000, 50), (7, 50), (1000, 40), (1000, 60)):
    batch = make_batch(4, 150000, seed, scale)
    idx = batch["voxel_locs"].int().to(dev)
    shape = [int(s) for s in batch["spatial_shape"]]
    sub = spconv.ops.build_subm(idx, 4, shape, 3)
    m = idx.shape[0]
    t = ext.with_tilebook(sub.tbl)
    nt, o64, o32 = ext.tilebook_overflow(t)
    tb = ops.tilebook_build(sub.tbl)
    c = 16
    x = torch.randn(m, c, device=dev).bfloat16(); z = torch.empty_like(x); y = torch.empty_like(x); r = torch.randn(m, c, device=dev).bfloat16()
    w = torch.randn(27, c, c, device=dev) * 0.05
    plan = ops.PackPlan([(w, 27, c, c, 0, 2)], dev); plan.run()
    vec = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    a = timed(lambda k: ops.spconv_gather(x, None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=r, want_stats=True, out=y))
    b = timed(lambda k: ops.spconv_gather(x, None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=r, want_stats=True, out=y, pre=(*vec, True, z)))
    print("seed %d scale %d: m %d tiles %d over64 %d over32 %d  plain %.1f us  prologue %.1f us" % (seed, scale, m, nt, o64, o32, a, b), flush=True)
