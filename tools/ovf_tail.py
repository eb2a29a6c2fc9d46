"""How much of conv_tile's time is the tail of its overflow tiles (tiles with more distinct neighbour rows than the list holds:
served from the dense table, a chain of dependent table -> row round trips)?  The bench scene's level-1 table as is, and
with the overflow tiles' outermost offsets removed until every tile fits its list (same rows, ~same pair count).
  python tools/ovf_tail.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import ops, spconv
from doda_amd._ext import ext
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")


def timed(fn, n=40):
    for k in range(5): fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
sub = spconv.ops.build_subm(idx, 4, shape, 3)
m = idx.shape[0]
tbl = sub.tbl.clone()
for c in (16, 32):
    x = torch.randn(m, c, device=dev).bfloat16(); y = torch.empty_like(x); r = torch.randn(m, c, device=dev).bfloat16()
    w = torch.randn(27, c, c, device=dev) * 0.05
    plan = ops.PackPlan([(w, 27, c, c, 0, 2)], dev); plan.run()
    vec = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    z = torch.empty_like(x)
    for trimmed in (0, 1):
        t = tbl.clone()
        if trimmed:
            cap = 1024 if c == 16 else 960
            for rnd in range(12):
                uc = ext.tilebook_parts(ext.with_tilebook(t))[2]
                over = (uc > cap).nonzero().flatten().tolist()
                if not over:
                    break
                for tile in over:   # drop the corner offsets of the tile's rows, a few more each round
                    for o in (0, 2, 6, 8, 18, 20, 24, 26, 1, 3, 5, 7)[: 4 + 2 * rnd]:
                        t[o, tile * 256:(tile + 1) * 256] = -1
        tw = ext.with_tilebook(t)
        nt, o64, o32 = ext.tilebook_overflow(tw)
        tb = ops.tilebook_build(t)
        a = timed(lambda k: ops.spconv_gather(x, None, t, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=r, want_stats=True, out=y))
        b = timed(lambda k: ops.spconv_gather(x, None, t, m, 0, c, packed=plan.outputs[0], tilebook=tb, residual=r, want_stats=True, out=y,
                                              pre=(*vec, True, z)))
        print("c=%d %s: tiles %d, above the 64-byte capacity %d, above the list %d, pairs %d: conv_tile %.1f us, with prologue %.1f us"
              % (c, "trimmed" if trimmed else "as is", nt, o64, o32, int((t >= 0).sum()), a, b), flush=True)
