"""How long the issuing thread waits for the helper thread's pyramid (future.result()) and spends handing it over
(PyramidPrefetcher.take) per step of the bench loop, next to the step's wall time; optional switch interval.
  python tools/takewait.py [steps] [switch_interval_seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.host import pin_to_device_numa
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if len(sys.argv) > 2:
    sys.setswitchinterval(float(sys.argv[2]))
pin_to_device_numa(0)
d = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
wp, wt = True, tile_levels_for(torch.bfloat16)
pf = PyramidPrefetcher(d, 7)
pend = [pf.submit(bd, wp, wt, resident=True, now=True)]
acc = [0.0, 0.0, 0.0, 0.0]
iv = []


def step(rec):
    a = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    pend[0].result()
    b = time.perf_counter()
    pyr = PyramidPrefetcher.take(pend[0], d)
    pend[0] = pf.submit(bd, wp, wt, resident=True)
    c = time.perf_counter()
    loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
    e = time.perf_counter()
    loss.backward()
    opt.step()
    f = time.perf_counter()
    if rec:
        iv.append(((f - a) * 1e3, (b - a) * 1e3, (c - b) * 1e3, (e - c) * 1e3, (f - e) * 1e3, len(iv)))
        acc[0] += b - a; acc[1] += c - b; acc[2] += e - c; acc[3] += f - e


for _ in range(20): step(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step(True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
print("switch interval %.4f s: wall %.2f ms/step | wait for the pyramid %.3f ms, take + submit %.3f ms, forward issue %.2f ms, backward + optimizer issue %.2f ms"
      % (sys.getswitchinterval(), wall, acc[0] / steps * 1e3, acc[1] / steps * 1e3, acc[2] / steps * 1e3, acc[3] / steps * 1e3))
import gc
print("gc counts", gc.get_count(), "collections per generation", [g["collections"] for g in gc.get_stats()])
iv.sort()
print("per-step issue interval ms: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (iv[0][0], iv[len(iv) // 10][0], iv[len(iv) // 2][0], iv[9 * len(iv) // 10][0], iv[-1][0]))
for r in iv[-8:]:
    print("   slow step %3d: total %.2f = wait %.2f + take %.2f + forward %.2f + backward/opt %.2f" % (r[5], r[0], r[1], r[2], r[3], r[4]))
pend[0].result(); pf.shutdown()
