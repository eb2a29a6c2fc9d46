#!/usr/bin/env python
"""What paces the training step?  One process, the bench batch (4 x 150k points), bf16:
  A  prefetch      rulebooks of the next batch on the helper thread + side stream (bench default)
  B  reuse         ONE pyramid reused by every step (diagnosis only: no rulebook work at all)
  C  inline        rulebooks built inside the step (side stream, main thread)
plus the wall time of one pyramid build on the helper thread while the step is running / idle.
Usage: stepdiag.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
from doda_amd.optim import FusedSGD
opt = (torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True) if os.environ.get("TORCH_SGD") == "1"
       else FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4))
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
PF = PyramidPrefetcher(dev, 7)
build_ms = []


def timed_build(*a):
    t = time.perf_counter(); r = PyramidPrefetcher._build(PF, *a); r[2].synchronize()
    build_ms.append((time.perf_counter() - t) * 1e3); return r


def run(mode, n):
    pend = [PF.submit(bd, wp)]
    fixed = PyramidPrefetcher.take(PF.submit(bd, wp), dev) if mode == "reuse" else None

    def step():
        opt.zero_grad(set_to_none=True)
        pyr = None
        if mode == "prefetch":
            pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(bd, wp)
        elif mode == "reuse":
            pyr = fixed
        s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr)
        l = cross_entropy(s, bd["labels"]); l.backward(); opt.step()
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    pend[0].result()
    return dt


for rnd in range(2):
    for mode in ("prefetch", "reuse", "inline"):
        print("round %d  %-9s %.3f ms/step" % (rnd, mode, run(mode, steps)), flush=True)

# pyramid build latency on the helper thread: GPU otherwise idle, then under a running step
torch.cuda.synchronize()
coords, shape, bs = bd["voxel_locs"], bd["spatial_shape"], bd["offsets"].numel() - 1
for _ in range(5):
    PF.pool.submit(timed_build, coords, shape, bs, wp).result()
print("pyramid build, idle GPU: %.2f ms (median of 5)" % sorted(build_ms)[2])
build_ms.clear()
fixed = PyramidPrefetcher.take(PF.submit(bd, wp), dev)
for k in range(12):
    f = PF.pool.submit(timed_build, coords, shape, bs, wp)
    opt.zero_grad(set_to_none=True)
    s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=fixed)
    l = cross_entropy(s, bd["labels"]); l.backward(); opt.step()
    f.result()
torch.cuda.synchronize()
print("pyramid build, under a running step: %.2f ms (median of 12)" % sorted(build_ms)[6])
PF.shutdown()
