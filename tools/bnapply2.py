"""The default BatchNorm apply sweeps alone — bn_apply behind doda_bn_relu_fwd_stats, bn_bwd_apply behind
doda_bn_relu_bwd_stats (statistics rows given: final + apply, two launches; the final launch is ~5 us of it) — at the U-Net's
level sizes: us per call, warm (same buffers) and cold (cycled through > 256 MB).  rocprofv3 --kernel-trace --stats of this
script gives the apply kernels alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from doda_amd import ops
from doda_amd._lib import lib, check
d = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for m, c in [(601279, 16), (601279, 32), (147000, 32), (147000, 64), (35000, 48), (8400, 64)]:
    nset = max(2, int(300e6 // (m * c * 2 * 4)) + 1)
    xs = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    ys = [torch.empty_like(x) for x in xs]
    dys = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    adds = [torch.randn(m, c, device=d).bfloat16() for _ in range(nset)]
    g = torch.rand(c, device=d) + 0.5; b = torch.randn(c, device=d) * 0.1
    rows = torch.rand(64, 2, c, device=d)
    mean = torch.empty(c, device=d); invstd = torch.empty(c, device=d)
    rm = torch.zeros(c, device=d); rv = torch.ones(c, device=d)

    def fwd(k):
        check(lib().doda_bn_relu_fwd_stats(xs[k].data_ptr(), m, c, 2, rows.data_ptr(), 64, 1e-4, 0.1, g.data_ptr(), b.data_ptr(),
                                           rm.data_ptr(), rv.data_ptr(), None, 1, ys[k].data_ptr(), mean.data_ptr(),
                                           invstd.data_ptr(), st()), "fwd")

    def bwd(k):
        ops.bn_relu_bwd_stats(xs[k], dys[k], rows, mean, invstd, g, b, True, adds[k])

    def timed(fn, cold, reps=60):
        for k in range(5):
            fn(k % nset if cold else 0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(reps):
            fn(k % nset if cold else 0)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for name, fn, nbytes in (("final + apply    ", fwd, m * c * 2 * 2), ("final + bwd_apply", bwd, m * c * 2 * 4)):
        w, cd = timed(fn, False), timed(fn, True)
        print("%7d x %3d %s warm %6.2f us | cold %6.2f us (%5.2f TB/s of the sweep's bytes)" % (m, c, name, w, cd, nbytes / cd / 1e6), flush=True)
