#!/usr/bin/env python
"""Debug variant of the tilebook builder (tools/_tbdbg/libdoda_hip.so, -DDODA_TB_DEBUG): after the hash path's sort every tile checks
that its list is strictly ascending; the first violation dumps the keys before and after the sort.  Builds the level-1 tilebook of the
Z-ordered 2 M-voxel batch N times and analyses the dump."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from doda_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_tbdbg", "libdoda_hip.so")
if not os.path.exists(_lib.LIB_PATH) or "--build" in sys.argv:      # (build it in the container: hipcc cross-compiles; the .so travels)
    import subprocess
    from doda_amd import build as B
    B.build_native(verbose=False)
    os.makedirs(os.path.dirname(_lib.LIB_PATH), exist_ok=True)
    obj = os.path.join(os.path.dirname(_lib.LIB_PATH), "tilebook.o")
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-DDODA_TB_DEBUG", "-c", os.path.join(B.HERE, "csrc", "tilebook.hip"), "-o", obj])
    objs = [os.path.join(B.OBJ, f) for f in sorted(os.listdir(B.OBJ)) if f.endswith(".o") and f != "tilebook.o"]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj, *objs, "-o", _lib.LIB_PATH])
    if "--build" in sys.argv:
        sys.exit(0)
import numpy as np, torch
from doda_amd import ops, spconv
from doda_amd.collate import reorder_voxels
from doda_amd.scene import make_batch
n_builds = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(4, 500000, 1000, 100), "morton")
idx = b["voxel_locs"].int().to(dev)
shape = [int(s) for s in b["spatial_shape"]]
tbl = ops.rulebook_subm(idx, shape, 4, 3)
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_uint32 * (8 + 4096))()
for r in range(n_builds):
    ops.tilebook_build(tbl)
    if r % 200 == 199 or r == n_builds - 1:
        torch.cuda.synchronize()
        lib.doda_tilebook_debug(buf, 0)
        if buf[0] > 0:
            break
a = np.frombuffer(buf, dtype=np.uint32)
print("builds run: %d, violations flagged: %d" % (r + 1, a[0]))
if a[0]:
    tile, U, P, viol = int(a[1]), int(a[2]), int(a[3]), int(a[4])
    pre, post = a[8:8 + 2048].copy(), a[8 + 2048:8 + 4096].copy()
    tc = tbl[:, tile * 256:(tile + 1) * 256].cpu().numpy()
    true = np.unique(tc[tc >= 0]).astype(np.uint32)
    pk = pre[:P]
    pre_keys = pk[pk != 0xFFFFFFFF]
    print("tile %d: U %d, P %d, %d order violations after the sort" % (tile, U, P, viol))
    print("per wave (P * 10000 + U):", a[8 + 4096 - 8:8 + 4096 - 4].tolist(), " sort stages counted per wave:", a[8 + 4096 - 4:8 + 4096].tolist())
    print("before the sort: %d keys in the first P, %d distinct, equal to the true set: %s; keys beyond P that are not EMPTY: %d" % (
        pre_keys.size, np.unique(pre_keys).size, np.array_equal(np.sort(pre_keys), true), int((pre[P:] != 0xFFFFFFFF).sum())))
    po = post[:U]
    print("after the sort: strictly ascending %s; multiset equal to the sorted input: %s" % (bool((np.diff(po.astype(np.int64)) > 0).all()), np.array_equal(np.sort(pre_keys), po)))
    bad = np.nonzero(np.diff(post[:P].astype(np.int64)) <= 0)[0]
    print("violations at positions", bad[:10].tolist())
    k0 = int(bad[0])
    print("post[%d:%d] =" % (k0 - 3, k0 + 6), post[max(0, k0 - 3):k0 + 6].tolist())
    print("sorted(pre) there =", np.sort(pre_keys)[max(0, k0 - 3):k0 + 6].tolist())
