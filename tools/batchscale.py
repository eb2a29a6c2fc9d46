#!/usr/bin/env python
"""Step time against batch size (scenes of ~150k voxels), one pyramid reused (no rulebook work), bf16:
the intercept of T(B) = a + b*B is the per-step cost that does not scale with rows (launch floors,
coarse levels, optimizer), the slope the throughput-bound part.  Usage: batchscale.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
from doda_amd.optim import FusedSGD
opt = (torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True) if os.environ.get("TORCH_SGD") == "1"
       else FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4))
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
PF = PyramidPrefetcher(dev, 7)
res = []
for B in [int(b) for b in os.environ.get("BS", "1,2,4,8,4,2,1").split(",")]:
    batch = make_batch(B, 150000, 1000)
    bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    fixed = PyramidPrefetcher.take(PF.submit(bd, wp), dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    seg = [0.0, 0.0, 0.0]

    def step(rec=False):
        opt.zero_grad(set_to_none=True)
        if rec: ev[0].record()
        s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=fixed)
        l = cross_entropy(s, bd["labels"])
        if rec: ev[1].record()
        l.backward()
        if rec: ev[2].record()
        opt.step()
        if rec: ev[3].record()
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    # GPU-side phase times (events in the stream; valid while the GPU is the bottleneck)
    for _ in range(5):
        step(True); torch.cuda.synchronize()
        for k in range(3): seg[k] += ev[k].elapsed_time(ev[k + 1]) / 5
    m = int(bd["voxel_locs"].shape[0])
    print("B %d  voxels %7d  %.3f ms/step  (%.1f Mvox/s)   isolated step on the GPU: fwd %.2f  bwd %.2f  opt %.2f ms"
          % (B, m, dt, m / dt / 1e3, *seg), flush=True)
    res.append((B, dt))
PF.shutdown()
