"""Ablations of conv_dma16 (DODA_DMA_DBG bits: 1 no multiply, 2 no row DMA, 4 no index-strip DMA, 8 no list loads):
plain forward, cold and warm.  One process per setting (the flag is read once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, torch
sys.path.insert(0, %r)
import bench
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
d = torch.device("cuda:0")
b = make_batch(int(sys.argv[1]), 150000, 1000)
idx = b["voxel_locs"].int().to(d)
data = spconv.ops.build_subm(idx, int(sys.argv[1]), b["spatial_shape"], 3)
m = idx.shape[0]
w = torch.randn(27, 16, 16, device=d) * 0.1
plan = ops.PackPlan([(w, 27, 16, 16, 0, 2)], d); plan.run(); pk = plan.outputs[0]
n = 6 if int(sys.argv[1]) > 1 else 16
xs = [torch.randn(m, 16, device=d).bfloat16() for _ in range(n)]
ys = [torch.empty(m, 16, device=d, dtype=torch.bfloat16) for _ in range(n)]
tbls = [data.tbl.clone() for _ in range(n)]
tbs = [ops.tilebook_build(t) for t in tbls]
k = [0]
def cold():
    j = k[0] = (k[0] + 1) %% n
    ops.spconv_gather(xs[j], None, tbls[j], m, 0, 16, packed=pk, tilebook=tbs[j], out=ys[j])
def warm():
    ops.spconv_gather(xs[0], None, tbls[0], m, 0, 16, packed=pk, tilebook=tbs[0], out=ys[0])
print(json.dumps({"cold_us": bench._timed(cold, 60) * 1e6, "warm_us": bench._timed(warm, 60) * 1e6}))
''' % ROOT
for scenes in (4, 1):
    for dbg in (0, 15, 31, 16, 17, 32, 47, -1):
        env = dict(os.environ)
        if dbg < 0:
            env["DODA_DMA_DBG"] = "0"; env["DODA_DMA"] = "0"
        else:
            env["DODA_DMA_DBG"] = str(dbg); env["DODA_DMA"] = "1"
        r = subprocess.run([sys.executable, "-c", CODE, str(scenes)], env=env, capture_output=True, text=True)
        print(scenes, dbg, r.stdout.strip() or r.stderr[-500:], flush=True)
