import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from doda_amd import model as M, spconv
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
from doda_amd._ext import ext
d = torch.device("cuda:0")
cfg = M.default_cfg(); torch.manual_seed(0)
net = M.SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 2000, 1000).items()}
pf = M.PyramidPrefetcher(d, 7)
pend = [pf.submit(bd, True, M.tile_levels_for(torch.bfloat16), resident=True, now=True)]
T = {"call": 0.0, "fc": 0.0, "n": 0}
orig = ext.coarse_ublock
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); T["call"] += time.perf_counter() - t; return r
class Shim:
    def __getattr__(self, n): return timed if n == "coarse_ublock" else getattr(ext, n)
spconv.functional._ext = Shim()
M.Fsp._ext = spconv.functional._ext
ofc = M.UBlock._forward_coarse
def fc(self, inp):
    t = time.perf_counter(); r = ofc(self, inp); T["fc"] += time.perf_counter() - t; T["n"] += 1; return r
M.UBlock._forward_coarse = fc
def step():
    opt.zero_grad(set_to_none=True)
    pyr = M.PyramidPrefetcher.take(pend[0], d)
    pend[0] = pf.submit(bd, True, M.tile_levels_for(torch.bfloat16), resident=True)
    a = time.perf_counter()
    loss = M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr, labels=bd["labels"])
    b = time.perf_counter()
    loss.backward(); c = time.perf_counter(); opt.step()
    return b - a, c - b
for _ in range(10): step()
T.update(call=0.0, fc=0.0, n=0); f = b = 0.0
for _ in range(50):
    x, y = step(); f += x; b += y
torch.cuda.synchronize()
print("per step: forward %.3f ms (of which _forward_coarse %.3f, of which the extension call %.3f), backward %.3f ms" % (f / 50 * 1e3, T["fc"] / 50 * 1e3, T["call"] / 50 * 1e3, b / 50 * 1e3))
pend[0].result(); pf.shutdown()
