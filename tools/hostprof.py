import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 1000)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
from doda_amd.optim import FusedSGD
opt = (torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True) if os.environ.get("TORCH_SGD") == "1"
       else FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4))
from doda_amd.spconv import functional as Fsp
print('deferred wgrad:', Fsp.set_deferred_wgrad(os.environ.get('DEFER', '1') == '1'))
from doda_amd.model import PyramidPrefetcher
from doda_amd import spconv
PF = PyramidPrefetcher(dev, 7) if os.environ.get('PREFETCH', '1') == '1' else None
wp = bool(spconv.functional.WGRAD_PAIRS)
pend = [PF.submit(bd, wp)] if PF else None
def fwd():
    opt.zero_grad(set_to_none=True)
    pyr = None
    if PF:
        pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(bd, wp)
    s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr)
    return cross_entropy(s, bd["labels"])
for _ in range(5):
    l = fwd(); l.backward(); opt.step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter(); tf = tb = to = 0.0
for _ in range(n):
    a = time.perf_counter(); l = fwd(); b = time.perf_counter(); l.backward(); c = time.perf_counter(); opt.step(); d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
torch.cuda.synchronize()
print("host issue time per step: fwd %.2f ms  bwd %.2f ms  opt %.2f ms ; wall %.2f ms" % (tf / n * 1e3, tb / n * 1e3, to / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    l = fwd(); l.backward(); opt.step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
