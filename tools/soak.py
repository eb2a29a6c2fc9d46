import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batches = []
for s in range(3):
    b = make_batch(4, 120000 + 20000 * s, 1000 + 10 * s)
    batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
PF = PyramidPrefetcher(dev, 7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
pend = [PF.submit(batches[0], wp)]
losses = []
t0 = time.perf_counter()
for k in range(n):
    bd = batches[k % 3]
    opt.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(batches[(k + 1) % 3], wp)
    s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr)
    l = cross_entropy(s, bd["labels"]); l.backward(); opt.step()
    if k % 250 == 0 or k == n - 1:
        losses.append(float(l))
        print("step %5d loss %.4f  allocated %.0f MB reserved %.0f MB  %.2f ms/step" % (
            k, losses[-1], torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
            (time.perf_counter() - t0) / (k + 1) * 1e3), flush=True)
pend[0].result(); PF.shutdown()
assert all(x == x and x < 10 for x in losses), losses
assert losses[-1] < losses[0]
print("soak ok")
