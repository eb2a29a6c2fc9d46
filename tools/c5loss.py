#!/usr/bin/env python
"""Loss curve of the bench step on a 1 cm batch (Z-order numbering), rulebooks prefetched as bench.py does: a check that the
curves of two settings (environment) agree.  usage: c5loss.py [steps=40] [scenes=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, 500000, 1000, 100), os.environ.get("ORDER", "morton"))
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
pf = PyramidPrefetcher(dev, 7)
pend = [pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True, now=True)]
out = []
for k in range(steps):
    opt.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pend[0], dev)
    pend[0] = pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True)
    loss = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"], ignore_index=255)
    loss.backward(); opt.step()
    if k % 4 == 0 or k == steps - 1:
        out.append("%d:%.4f" % (k, float(loss.detach())))
pend[0].result(); pf.shutdown()
print(" ".join(out))
