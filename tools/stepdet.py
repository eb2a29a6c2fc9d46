#!/usr/bin/env python
"""Is one training step bit-reproducible?  Same weights, same batch, R repetitions of forward + backward (rulebooks prefetched on
the helper thread as bench.py does): loss and every parameter gradient compared bitwise with the first repetition.
usage: stepdet.py [reps=8] [scenes=4] [voxel_scale=100] [voxels=500000]   (environment switches select the variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.scene import make_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
nv = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, nv, 1000, vs), os.environ.get("ORDER", "morton"))
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
pf = PyramidPrefetcher(dev, 7) if os.environ.get("PREFETCH", "1") == "1" else None
state = {k: v.clone() for k, v in net.state_dict().items()}
first, bad = None, 0
for r in range(reps):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True, now=True), dev) if pf else None
    loss = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"], ignore_index=255)
    loss.backward()
    torch.cuda.synchronize()
    cur = (loss.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()})
    if first is None:
        first = cur
        continue
    diff = [k for k in cur[1] if not torch.equal(cur[1][k], first[1][k])]
    if diff or not torch.equal(cur[0], first[0]):
        bad += 1
        worst = max(((float((cur[1][k].float() - first[1][k].float()).norm() / (first[1][k].float().norm() + 1e-30)), k) for k in diff), default=(0.0, ""))
        print("rep %d: loss %s (first %s), %d of %d gradients differ; worst %.3e %s; e.g. %s" % (
            r, float(cur[0]), float(first[0]), len(diff), len(cur[1]), worst[0], worst[1], diff[:4]), flush=True)
if pf:
    pf.shutdown()
print("repetitions that differ from the first: %d of %d" % (bad, reps - 1))
