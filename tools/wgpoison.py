#!/usr/bin/env python
"""Does any weight-gradient kernel read workspace it did not write?  Every job class of doda_spconv_wgrad_multi on the 1 cm B4
batch (2.0 M voxels, Z-order numbering), its workspace and output poisoned with NaN bit patterns before each call, three runs:
results must be finite and identical.  usage: wgpoison.py [scenes=4] [voxel_scale=100] [voxels=500000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import ops, spconv
from doda_amd.collate import reorder_voxels
from doda_amd.scene import make_batch
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, nv, 1000, vs), "morton")
idx = b["voxel_locs"].int().to(dev)
shape = [int(s) for s in b["spatial_shape"]]
t = spconv.SparseConvTensor(None, idx, shape, ns)
books = spconv.ops.build_pyramid(t, 4, with_pairs=False, with_tiles=2)
g = torch.Generator().manual_seed(0)


def feats(n, c):
    return torch.randn(n, c, generator=g).bfloat16().to(dev)


def poison():
    for buf in ops._WS.values():
        buf.fill_(0xFF)


cases = []
for lvl, c in ((1, 16), (2, 32), (3, 48), (4, 64)):
    sub = books["subm%d" % lvl]
    n = sub.tbl.shape[1]
    x, dy = feats(n, c), feats(n, c)
    has_tb = spconv.ops._ext.has_tilebook(sub.tbl)
    tb = ops.tilebook_build(sub.tbl) if has_tb else None
    cases.append(("subm%d %d->%d table" % (lvl, c, c), [(x, dy, sub.tbl, n)]))
    if tb is not None:
        cases.append(("subm%d %d->%d tile" % (lvl, c, c), [(x, dy, sub.tbl, n, None, None, tb)]))
        cases.append(("subm%d %d->%d tile x5" % (lvl, c, c), [(x, dy, sub.tbl, n, None, None, tb)] * 5))
    ident = torch.arange(n, dtype=torch.int32, device=dev).view(1, n)
    cases.append(("1x1 level %d %d->%d table" % (lvl, c, c), [(x, dy, ident, n)]))
    if c % 16 == 0:
        cases.append(("1x1 level %d %d->%d pairs" % (lvl, c, c), [(x, dy, None, n, (ident, ident, None))]))
    if lvl < 4:
        dn = books["spconv%d" % lvl]
        m_out = dn.outids.shape[0]
        co = {1: 32, 2: 48, 3: 64}[lvl]
        cases.append(("down %d->%d (%d->%d rows) table" % (c, co, n, m_out), [(x, feats(m_out, co), dn.tbl, m_out)]))
        cases.append(("up %d->%d table" % (co, c), [(feats(m_out, co), dy, dn.tbl_rev, n)]))
# the step issues MANY jobs per call: the classes together, as the deferred flush does
table_jobs = [j for name, jobs in cases if "table" in name for j in jobs]
cases.append(("all table-class jobs in ONE call (%d)" % len(table_jobs), table_jobs))
all_jobs = [j for name, jobs in cases[:-1] if "x5" not in name for j in jobs]
cases.append(("every job in ONE call (%d)" % len(all_jobs), all_jobs))
# the decoder's 1x1 skip convs change the channel count (unet_block.py:18-21): 32 -> 16 at level 1, 64 -> 32 at level 2
for lvl, ci, co in ((1, 32, 16), (2, 64, 32), (3, 96, 48)):
    n = books["subm%d" % lvl].tbl.shape[1]
    ident = torch.arange(n, dtype=torch.int32, device=dev).view(1, n)
    x, dy = feats(n, ci), feats(n, co)
    cases.append(("1x1 level %d %d->%d table" % (lvl, ci, co), [(x, dy, ident, n)]))
    cases.append(("1x1 level %d %d->%d pairs" % (lvl, ci, co), [(x, dy, None, n, (ident, ident, None))]))
    cases.append(("subm%d %d->%d table" % (lvl, ci, co), [(x, dy, books["subm%d" % lvl].tbl, n)]))
bad = 0
for name, jobs in cases:
    outs = []
    for rep in range(3):
        poison()
        torch.cuda.synchronize()
        dws = ops.spconv_wgrad_multi(jobs)
        torch.cuda.synchronize()
        outs.append([d.clone() for d in dws])
    finite = all(bool(torch.isfinite(d).all()) for d in outs[0])
    same = all(torch.equal(a, c) for r in outs[1:] for a, c in zip(outs[0], r))
    if not (finite and same):
        bad += 1
    print("%-48s finite %s  identical over 3 runs %s" % (name, finite, same), flush=True)
print("cases with a problem:", bad)
