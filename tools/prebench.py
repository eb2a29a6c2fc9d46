#!/usr/bin/env python
"""Micro-benchmark of the folded BatchNorm (doda_conv_prologue) against the two launches it replaces, at one U-Net level's size.
usage: python tools/prebench.py [rows channels] ...   (default: levels 4-7 of the bench batch)
Per configuration, microseconds per launch, back to back on one stream (kernel + boundary): the plain SubM conv with statistics,
the standalone BatchNorm sweeps of the per-layer backend, the folded forward (kind 1) and backward (kinds 2 / 3) convs.
A library built with DODA_EXTRA_HIPCC_FLAGS=-DDODA_PRE_ABLATE reads DODA_PRE_ABLATE (1 no side sweep, 2 no transform, 4 no totals)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from doda_amd import ops
from tests.util import surface_voxels

d = torch.device("cuda:0")
args = [int(a) for a in sys.argv[1:]]
cfgs = list(zip(args[0::2], args[1::2])) or [(8400, 64), (1900, 80), (420, 96), (83, 112)]
BIG = 1 << 30


def timed(lst, reps=200, calls=4):
    """us per repetition of `lst`: the list is repeated inside ONE doda_layers_run call (no interpreter between the launches)."""
    big = lst * reps
    ops.layers_run(big, d, 2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(calls):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.layers_run(big, d, 2)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for n, c in cfgs:
    batch = 4
    side = max(16, int(round((n / batch / 0.08) ** (1 / 3))))
    shape = [side] * 3
    idx = np.ascontiguousarray(surface_voxels(n, n, batch, shape)[:n])
    n = idx.shape[0]
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    g = torch.Generator().manual_seed(n)
    bf = lambda t: t.to(torch.bfloat16).to(d)
    x, dy, add = bf(torch.randn(n, c, generator=g)), bf(torch.randn(n, c, generator=g)), bf(torch.randn(n, c, generator=g))
    a, y, du = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    w = (torch.randn(27, c, c, generator=g) * (1.0 / (c * 9)) ** 0.5).to(d)
    plan = ops.PackPlan([(w, 27, c, c, 0, 2), (w, 27, c, c, 2, 2)], d)
    plan.run()
    wf, wb = plan.outputs
    gamma, beta = torch.ones(c, device=d), torch.zeros(c, device=d)
    mean, invstd = x.float().mean(0), 1.0 / torch.sqrt(x.float().var(0, unbiased=False) + 1e-4)
    rm, rv, nbt = torch.zeros(c, device=d), torch.ones(c, device=d), torch.zeros(1, dtype=torch.int64, device=d)
    tx, ty, tb = ops.stats_totals(c, d), ops.stats_totals(c, d), ops.stats_totals(c, d)
    dg, db = torch.zeros(c, device=d), torch.zeros(c, device=d)
    ops.layers_run([dict(kind=ops.CX_STATS, flags=0, rows=n, c_in=c, x_ld=c, x=x, stats=tx)], d, 2)
    bnf = dict(kind=ops.CX_BNFWD, flags=ops.CX_F_RELU | ops.CX_F_TRAINING, rows=n, c_in=c, x_ld=c, y_ld=c, x=x, y=a, eps=1e-4, momentum=0.1,
               gamma=gamma, beta=beta, running_mean=rm, running_var=rv, nbt=nbt, mean=mean.clone(), invstd=invstd.clone(), stats=tx, c_split=c)
    gf = dict(kind=ops.CX_GEMM, flags=0, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, x=a, w=wf, tbl=tbl, y=y, stats=ty)
    gplain = dict(gf, stats=None)
    bnb = dict(kind=ops.CX_BNBWD, flags=ops.CX_F_RELU, rows=n, c_in=c, c_split=c, x_ld=c, y_ld=c, aux_ld=c, x=dy, aux=x, y=du, stats=tb,
               mean=mean, invstd=invstd, gamma=gamma, beta=beta, dgamma=dg, dbeta=db)
    bnb3 = dict(bnb, res=add, res_ld=c)
    gb = dict(kind=ops.CX_GEMM, flags=ops.CX_F_RELU, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, x=du, w=wb, tbl=tbl, y=y,
              aux=x, aux_ld=c, mean=mean, invstd=invstd, gamma=gamma, beta=beta, stats=ty)
    res = {}
    old = ops.set_pre_rows(0, 0)
    res["conv"] = timed([gplain]); res["conv+stats"] = timed([gf]); res["dgrad+bnstats"] = timed([gb])
    res["bn fwd"] = timed([bnf]); res["bn bwd"] = timed([bnb]); res["bn bwd+add"] = timed([bnb3])
    ops.set_pre_rows(BIG, BIG)
    res["fold fwd"] = timed([bnf, gf]); res["fold bwd"] = timed([bnb, gb]); res["fold bwd+add"] = timed([bnb3, gb])
    ops.set_pre_rows(*old)
    print("rows %6d c %3d : " % (n, c) + "  ".join("%s %.1f" % (k, v) for k, v in res.items()), flush=True)
