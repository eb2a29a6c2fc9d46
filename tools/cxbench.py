"""Per-op times of the coarse-level executor: one launch of N copies of an op (barrier in front of each), HIP events."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from doda_amd import ops
from tests.test_gpu_coarse import _level, _pack, _bf

d = torch.device("cuda:0")
REP = 40
B = ops.CX_F_BARRIER


def timed(chain, reps=5):
    ops.coarse_run(chain, d)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.coarse_run(chain, d)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best


G = ops.coarse_workgroups()
print("workgroups", G)
for n, c in ((1900, 80), (420, 96), (83, 112), (8400, 64)):
    idx, shape, batch = _level(n, n)
    n = idx.shape[0]
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    pairs = int((tbl >= 0).sum())
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(n, c, generator=g)).to(d)
    y = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
    y2 = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
    w = (torch.randn(27, c, c, generator=g) * 0.05).to(d)
    wp = _pack(w, 27, c, c, 0, d)
    st = torch.zeros((G, 2, c), dtype=torch.float32, device=d)
    ga, be = torch.ones(c, device=d), torch.zeros(c, device=d)
    mean, invstd = torch.zeros(c, device=d), torch.ones(c, device=d)
    gemm = dict(kind=ops.CX_GEMM, flags=B, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, x=x, w=wp, tbl=tbl, y=y, stats=st)
    gemm_b = dict(gemm, flags=B | ops.CX_F_RELU, aux=x, aux_ld=c, mean=mean, invstd=invstd, gamma=ga, beta=be, y=y2)
    bnf = dict(kind=ops.CX_BNFWD, flags=B | ops.CX_F_RELU | ops.CX_F_TRAINING, rows=n, c_in=c, x_ld=c, y_ld=c, c_split=c, eps=1e-4, momentum=0.1,
               x=x, y=y2, stats=st, gamma=ga, beta=be, mean=mean, invstd=invstd)
    bnb = dict(kind=ops.CX_BNBWD, flags=B, rows=n, c_in=c, x_ld=c, aux_ld=c, y_ld=c, c_split=c, x=x, aux=x, y=y2, stats=st, mean=mean, invstd=invstd, gamma=ga)
    sts = dict(kind=ops.CX_STATS, flags=B, rows=n, c_in=c, x_ld=c, x=x, stats=st)
    t1 = timed([sts])
    print("rows %5d x %3d ch, %.1f pairs/row: launch of one op %.1f us" % (n, c, pairs / n, t1))
    for name, op in (("STATS", sts), ("BNFWD", bnf), ("BNBWD", bnb), ("GEMM fwd", gemm), ("GEMM bwd", gemm_b)):
        t = timed([op] * REP)
        print("   %-9s %7.2f us per op" % (name, (t - t1) / (REP - 1)))
