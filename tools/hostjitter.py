#!/usr/bin/env python
"""Distribution of the per-step HOST issue time (no sync inside the loop), for several scene sizes and with
the cyclic GC on / off.  Usage: hostjitter.py [steps]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
from doda_amd.optim import FusedSGD
opt = (torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True) if os.environ.get("TORCH_SGD") == "1"
       else FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4))
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
PF = PyramidPrefetcher(dev, 7)
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
for B, pts in ((4, 20000), (1, 150000), (4, 150000)):
    batch = make_batch(B, pts, 1000)
    bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    fixed = PyramidPrefetcher.take(PF.submit(bd, wp), dev)
    for use_gc in (True, False):
        gc.enable() if use_gc else gc.disable()
        ph = [[], [], []]

        def step():
            a = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            s = voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=fixed)
            l = cross_entropy(s, bd["labels"])
            b = time.perf_counter(); l.backward(); c = time.perf_counter(); opt.step(); d = time.perf_counter()
            ph[0].append(b - a); ph[1].append(c - b); ph[2].append(d - c)
        for _ in range(6): step()
        torch.cuda.synchronize()
        for p in ph: p.clear()
        t0 = time.perf_counter()
        for _ in range(steps): step()
        host = (time.perf_counter() - t0) / steps * 1e3
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / steps * 1e3
        med = [sorted(p)[len(p) // 2] * 1e3 for p in ph]
        mx = [max(p) * 1e3 for p in ph]
        print("B %d x %6d pts  gc %d : host %.2f ms/step, with final sync %.2f | median fwd %.2f bwd %.2f opt %.2f | max fwd %.2f bwd %.2f opt %.2f"
              % (B, pts, use_gc, host, wall, *med, *mx), flush=True)
gc.enable()
PF.shutdown()
