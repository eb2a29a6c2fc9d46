"""In-process A/B of step time: alternates configurations inside ONE process on ONE box (boxes and
processes differ by +-5 %, more than most of the effects being measured).
  python tools/abbench.py [steps]      -> ms/step per configuration, several rounds"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch
from doda_amd.spconv import functional as Fsp
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
Fsp.set_deferred_wgrad(True)
PF = PyramidPrefetcher(dev, 7)
def run(n, pairs, fusion):
    Fsp.WGRAD_PAIRS = pairs; Fsp.set_bn_fusion(fusion)
    pend = [PF.submit(bd, pairs)]
    def step():
        opt.zero_grad(set_to_none=True)
        pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(bd, pairs)
        l = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
        l.backward(); opt.step()
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    pend[0].result()
    return dt
for rnd in range(3):
    print("round %d: " % rnd + "  ".join("pairs=%d fusion=%d %.2f ms" % (p, f, run(steps, bool(p), bool(f))) for p, f in ((1, 1), (0, 1), (1, 0), (0, 0))), flush=True)
