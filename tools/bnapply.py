"""The BatchNorm apply sweeps alone (bn_apply_tot / bn_bwd_apply_tot through the totals entry points: one launch each) at
the U-Net's level sizes: us and GB/s, same buffers back to back (warm) and cycled through > 256 MB (cold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import ops
d = torch.device("cuda:0")
for m, c in [(601279, 16), (601279, 32), (150000, 32), (150000, 64), (40000, 48)]:
    nset = max(2, int(300e6 // (m * c * 2 * 4)) + 1)
    xs = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    dys = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    adds = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    g = torch.rand(c, device=d) + 0.5; b = torch.randn(c, device=d) * 0.1
    tot = torch.stack([xs[0].double().sum(0), (xs[0].double() ** 2).sum(0)]).contiguous()
    y, mean, invstd = ops.bn_relu_fwd_totals(xs[0], tot, g, b, None, None, 0.1, 1e-4, True)
    totb = torch.rand(2, c, device=d, dtype=torch.float64)

    def timed(fn, cold, reps=60):
        for k in range(5):
            fn(k % nset if cold else 0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(reps):
            fn(k % nset if cold else 0)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    fwd = lambda k: ops.bn_relu_fwd_totals(xs[k], tot, g, b, None, None, 0.1, 1e-4, True)
    bwd = lambda k: ops.bn_relu_bwd_totals(xs[k], dys[k], totb, mean, invstd, g, b, True, add=adds[k])
    for name, fn, nbytes in (("apply    ", fwd, m * c * 2 * 2), ("bwd_apply", bwd, m * c * 2 * 4)):
        w, cd = timed(fn, False), timed(fn, True)
        print("%7d x %3d %s warm %6.2f us %5.2f TB/s | cold %6.2f us %5.2f TB/s" % (m, c, name, w, nbytes / w / 1e6, cd, nbytes / cd / 1e6))
