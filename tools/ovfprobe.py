"""How many tiles of the bench batch exceed the tilebook's list capacity, and what the gate kernels cost with the
overflow paths forced off (DODA_DMA_DBG=64: wrong results on those tiles, timing only)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, torch
sys.path.insert(0, %r)
import bench
from doda_amd import ops, spconv
from doda_amd._lib import lib
from doda_amd.scene import make_batch
d = torch.device("cuda:0")
ns = int(sys.argv[1])
b = make_batch(ns, 150000, 1000)
idx = b["voxel_locs"].int().to(d)
data = spconv.ops.build_subm(idx, ns, b["spatial_shape"], 3)
m = idx.shape[0]
tb = ops.tilebook_build(data.tbl)
nt = (m + 255) // 256
uc = tb[nt * (lib().doda_tilebook_umax() * 4 + 27 * 512):].view(torch.int32)[:nt].cpu()
hist = {k: int((uc > k).sum()) for k in (768, 896, 960, 1024, 1152, 1280)}
w = torch.randn(27, 16, 16, device=d) * 0.1
plan = ops.PackPlan([(w, 27, 16, 16, 0, 2)], d); plan.run(); pk = plan.outputs[0]
n = 6 if ns > 1 else 16
xs = [torch.randn(m, 16, device=d).bfloat16() for _ in range(n)]
gs = [torch.randn(m, 16, device=d).bfloat16() for _ in range(n)]
ys = [torch.empty(m, 16, device=d, dtype=torch.bfloat16) for _ in range(n)]
tbls = [data.tbl.clone() for _ in range(n)]
tbs = [ops.tilebook_build(t) for t in tbls]
k = [0]
def cold():
    j = k[0] = (k[0] + 1) %% n
    ops.spconv_gather(xs[j], None, tbls[j], m, 0, 16, packed=pk, tilebook=tbs[j], out=ys[j])
jobs = [(xs[j %% n], gs[j %% n], tbls[j %% n], m, None, None, tbs[j %% n]) for j in range(8)]
print(json.dumps({"tiles": nt, "tiles_over": hist, "mean": float(uc.float().mean()), "max": int(uc.max()),
                  "conv_cold_us": bench._timed(cold, 60) * 1e6,
                  "wgrad_cold_us_per_layer": bench._timed(lambda: ops.spconv_wgrad_multi(jobs), 12, per=8) * 1e6}))
''' % ROOT
for scenes in (4, 1):
    for dbg in (0, 64):
        env = dict(os.environ, DODA_DMA_DBG=str(dbg))
        r = subprocess.run([sys.executable, "-c", CODE, str(scenes)], env=env, capture_output=True, text=True)
        print(scenes, dbg, r.stdout.strip() or r.stderr[-800:], flush=True)
