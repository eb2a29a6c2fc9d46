import cProfile, pstats, io, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from doda_amd import model as M, spconv
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
d = torch.device("cuda:0")
cfg = M.default_cfg(); torch.manual_seed(0)
net = M.SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 20000, 1000).items()}
pf = M.PyramidPrefetcher(d, 7)
pyr = M.PyramidPrefetcher.take(pf.submit(bd, True, M.tile_levels_for(torch.bfloat16), resident=True, now=True), d)
def fwd():
    return M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
def step():
    opt.zero_grad(set_to_none=True); l = fwd(); l.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
n = 20
for _ in range(n):
    opt.zero_grad(set_to_none=True)
    pr.enable(); l = fwd(); pr.disable()
    l.backward(); opt.step()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
s = io.StringIO(); st.stream = s; st.print_stats(45); out = s.getvalue()
print("\n".join(l[:170] for l in out.split("\n")[:75]))
pf.shutdown()
