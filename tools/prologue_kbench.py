#!/usr/bin/env python
"""conv_tile with and without the BatchNorm prologue, alone: level-1 (16 -> 16) and level-2-like (32 -> 32) SubM forward in
the form the step launches it (statistics + residual), warm (same operands back to back) and cold (operands cycled
through NSET buffer sets), HIP-event timed.  Also the standalone apply pass the prologue replaces.
  python tools/prologue_kbench.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
sub = spconv.ops.build_subm(idx, 4, shape, 3)
m = idx.shape[0]
tb = ops.tilebook_build(sub.tbl)
NSET = 6


def timed(fn, n):
    for k in range(3): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for c in (16, 32):
    xs = [torch.randn(m, c, device=dev).bfloat16() for _ in range(NSET)]
    zs = [torch.empty_like(x) for x in xs]
    ys = [torch.empty(m, c, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    rs = [torch.randn(m, c, device=dev).bfloat16() for _ in range(NSET)]
    w = torch.randn(27, c, c, device=dev) * 0.05
    plan = ops.PackPlan([(w, 27, c, c, 0, 2)], dev); plan.run()
    vec = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    for cold in (0, 1):
        sel = (lambda k: k % NSET) if cold else (lambda k: 0)
        plain = timed(lambda k: ops.spconv_gather(xs[sel(k)], None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb,
                                                  residual=rs[sel(k)], want_stats=True, out=ys[sel(k)]), reps)
        pre = timed(lambda k: ops.spconv_gather(xs[sel(k)], None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb,
                                                residual=rs[sel(k)], want_stats=True, out=ys[sel(k)],
                                                pre=(*vec, True, zs[sel(k)])), reps)
        pre_noz = timed(lambda k: ops.spconv_gather(xs[sel(k)], None, sub.tbl, m, 0, c, packed=plan.outputs[0], tilebook=tb,
                                                    residual=rs[sel(k)], want_stats=True, out=ys[sel(k)],
                                                    pre=(*vec, True, None)), reps)
        print("c=%d %s: conv_tile %.1f us, with prologue %.1f us, prologue without side output %.1f us"
              % (c, "cold" if cold else "warm", plain, pre, pre_noz), flush=True)
