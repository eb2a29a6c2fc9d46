#!/usr/bin/env python
"""Which operator of the forward pass comes out different?  N forward passes (training mode, same weights, same batch, rulebooks
built once) under DODA_FPLOG=1: every operator output of the extension is fingerprinted; a pass whose list differs from the first
pass's is reported with the FIRST differing operator.  usage: DODA_FPLOG=1 fwddet.py [passes=2000] [scenes=4] [scale=100] [voxels=500000]
BACKWARD=1: forward + backward per pass (the backward operators are logged too)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.scene import make_batch
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
nv = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
bwd = os.environ.get("BACKWARD", "0") == "1"
dev = torch.device("cuda:0")
ext = spconv.functional._ext
assert os.environ.get("DODA_FPLOG") == "1" and hasattr(ext, "fp_log_take")
b = reorder_voxels(make_batch(ns, nv, 1000, vs), os.environ.get("ORDER", "morton"))
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
spconv.functional.set_deferred_wgrad(True)
pf = PyramidPrefetcher(dev, 7)
wp = bool(spconv.functional.WGRAD_PAIRS)
state = {k: v.clone() for k, v in net.state_dict().items()}
ref, bad = None, 0
for r in range(passes):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True, now=True), dev)
    ext.fp_log_take()
    loss = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"], ignore_index=255)
    if bwd:
        loss.backward()
    lv = float(loss.detach())
    log = ext.fp_log_take()
    if ref is None:
        ref = (log, lv)
        print("operators logged per pass: %d, loss %.6f" % (len(log), lv), flush=True)
        continue
    if log != ref[0] or lv != ref[1]:
        bad += 1
        k = next((i for i in range(min(len(log), len(ref[0]))) if log[i] != ref[0][i]), None)
        if bad <= 12:
            print("pass %d: loss %.6f (first %.6f); first differing operator #%s: %s  (fingerprint %s vs %s); %d entries differ" % (
                r, lv, ref[1], k, log[k][0] if k is not None else None, log[k][1] if k is not None else None,
                ref[0][k][1] if k is not None else None, sum(1 for x, y in zip(log, ref[0]) if x != y)), flush=True)
            if k is not None and k > 0:
                print("      previous operator: %s" % (log[k - 1][0],), flush=True)
pf.shutdown()
print("passes that differ from the first: %d of %d" % (bad, passes - 1))
