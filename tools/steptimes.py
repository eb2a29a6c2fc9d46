#!/usr/bin/env python
"""Per-step times of the first 80 training steps after process start (HIP events at step boundaries, read
back at the end): how many steps does the bench need before it is in steady state?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
PF = PyramidPrefetcher(dev, 7)
pend = [PF.submit(bd, wp)]
n = 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize(); ev[0].record()
for k in range(n):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(bd, wp)
    s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr)
    l = cross_entropy(s, bd["labels"]); l.backward(); opt.step()
    ev[k + 1].record(); host.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
gpu = [ev[k].elapsed_time(ev[k + 1]) for k in range(n)]
for k0 in range(0, n, 10):
    print("steps %2d-%2d  gpu-side interval ms: %s" % (k0, k0 + 9, " ".join("%.2f" % v for v in gpu[k0:k0 + 10])))
    print("             host issue ms:        %s" % " ".join("%.2f" % v for v in host[k0:k0 + 10]))
pend[0].result(); PF.shutdown()
