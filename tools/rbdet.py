#!/usr/bin/env python
"""Are the rulebook builders deterministic at 2 M voxels?  The whole pyramid (13 rulebooks + tilebooks) built N times from the same
coordinates; every table compared bitwise with the first build's.  usage: rbdet.py [builds=1500] [scenes=4] [scale=100] [voxels=500000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.scene import make_batch
builds = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
nv = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, nv, 1000, vs), os.environ.get("ORDER", "morton"))
idx = b["voxel_locs"].int().to(dev)
shape = [int(s) for s in b["spatial_shape"]]


def build():
    t = spconv.SparseConvTensor(None, idx, shape, ns)
    books = spconv.ops.build_pyramid(t, 7, with_pairs=False, with_tiles=2)
    out = {}
    for key, d in books.items():
        out[key + ".tbl"] = d.tbl.clone()
        st = d.tbl.untyped_storage()
        nb = d.tbl.numel() * 4
        if st.nbytes() > nb:                                   # the tilebook behind the table (past the alignment gap)
            raw = torch.empty(0, dtype=torch.uint8, device=dev).set_(st)
            out[key + ".tilebook"] = raw[(nb + 255) // 256 * 256:].clone()
        if d.tbl_rev is not None:
            out[key + ".tbl_rev"] = d.tbl_rev.clone()
        out[key + ".outids"] = d.outids.clone()
    return out


ref = build()
torch.cuda.synchronize()
bad = 0
for r in range(builds):
    cur = build()
    diff = [k for k in ref if cur[k].shape != ref[k].shape or not torch.equal(cur[k], ref[k])]
    if diff:
        bad += 1
        if bad <= 10:
            k = diff[0]
            n = int((cur[k] != ref[k]).sum()) if cur[k].shape == ref[k].shape else -1
            pos = (cur[k].reshape(-1) != ref[k].reshape(-1)).nonzero().flatten()[:6].tolist() if n > 0 else []
            print("build %d: %d tensors differ: %s; %s: %d elements, first at %s (of %d)" % (r, len(diff), diff[:6], k, n, pos, ref[k].numel()), flush=True)
        if bad <= 4 and "subm1.tilebook" in diff:
            import numpy as np
            m = ref["subm1.tbl"].shape[1]
            nt = (m + 255) // 256
            a = ref["subm1.tilebook"][:nt * 4096].view(torch.int32).view(nt, 1024).cpu().numpy()
            c = cur["subm1.tilebook"][:nt * 4096].view(torch.int32).view(nt, 1024).cpu().numpy()
            e = np.arange(1024)
            upos = (((e >> 5) & 7) * 32 + (e & 31)) * 4 + (e >> 8)          # tilebook.hpp tb_upos
            tbl = ref["subm1.tbl"].cpu().numpy()
            for t in np.nonzero((a != c).any(1))[0][:3]:
                la, lc = a[t][upos], c[t][upos]
                ua, uc = int((la >= 0).sum()), int((lc >= 0).sum())
                v = tbl[:, t * 256:(t + 1) * 256]
                v = v[v >= 0]
                true_set = np.unique(v)
                print("   tile %d: distinct rows %d, span %d; list A: %d entries, sorted %s, equals the true set %s; list B: %d entries, sorted %s, equals the true set %s; first differing entry %d" % (
                    t, true_set.size, int(v.max()) - int(v.min()) + 1, ua, bool((np.diff(la[:ua]) > 0).all()), np.array_equal(la[:ua], true_set),
                    uc, bool((np.diff(lc[:uc]) > 0).all()), np.array_equal(lc[:uc], true_set), int(np.nonzero(la != lc)[0][0])), flush=True)
                k0 = int(np.nonzero(la != lc)[0][0])
                print("      A[%d:%d] = %s\n      B[%d:%d] = %s" % (k0 - 2, k0 + 6, la[max(0, k0 - 2):k0 + 6].tolist(), k0 - 2, k0 + 6, lc[max(0, k0 - 2):k0 + 6].tolist()), flush=True)
print("builds that differ from the first: %d of %d" % (bad, builds))
