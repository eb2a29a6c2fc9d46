#!/usr/bin/env python
"""Is a short training run bit-reproducible?  Same initial weights, same batch, R runs of S optimizer steps with the bench's step
(rulebooks of the next step prefetched while the current one runs, deferred weight gradients, FusedSGD): the final loss and a
checksum of all parameters must be identical.  usage: traindet.py [runs=4] [steps=12] [scenes=4] [voxel_scale=100] [voxels=500000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.collate import reorder_voxels
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
vs = int(sys.argv[4]) if len(sys.argv) > 4 else 100
nv = int(sys.argv[5]) if len(sys.argv) > 5 else 500000
dev = torch.device("cuda:0")
b = reorder_voxels(make_batch(ns, nv, 1000, vs), os.environ.get("ORDER", "morton"))
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
cfg = default_cfg()
spconv.functional.set_deferred_wgrad(True)
wp = bool(spconv.functional.WGRAD_PAIRS)
use_pf = os.environ.get("PREFETCH", "1") == "1"
inline = os.environ.get("INLINE", "0") == "1"      # rulebooks built in the forward pass, on the step's own stream
if inline:
    use_pf = False
res = []
for r in range(runs):
    torch.manual_seed(0)
    net = SparseConvNet(cfg).to(dev).train()
    opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    pf = PyramidPrefetcher(dev, 7) if use_pf else None
    pend = [pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True, now=True)] if pf else None
    losses = []
    for k in range(steps):
        opt.zero_grad(set_to_none=True)
        pyr = None
        if pf:
            pyr = PyramidPrefetcher.take(pend[0], dev)
            pend[0] = pf.submit(bd, wp, tile_levels_for(torch.bfloat16), resident=True)
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=not inline, pyramid=pyr), bd["labels"], ignore_index=255)
        loss.backward()
        if os.environ.get("GRADCHECK", "0") == "1":     # which parameter's gradient is off, and at which step
            names = [n for n, p in net.named_parameters()]
            norms = torch.stack(torch._foreach_norm([p.grad.float() for p in net.parameters()])).cpu()
            if r == 0:
                ref_norms = globals().setdefault("REF", {})
                ref_norms[k] = norms
            else:
                ref = globals()["REF"].get(k)
                if ref is not None and not torch.equal(ref, norms) and not globals().get("REPORTED_%d" % r):
                    globals()["REPORTED_%d" % r] = True
                    offs = [(names[j], float(ref[j]), float(norms[j])) for j in range(len(names)) if ref[j] != norms[j]]
                    offs.sort(key=lambda t: -abs(t[2] - t[1]) / (abs(t[1]) + 1e-30))
                    print("run %d step %d: %d gradient norms differ from run 0; largest relative deviations: %s" % (
                        r, k, len(offs), ["%s %.6g -> %.6g" % o for o in offs[:5]]), flush=True)
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    if pf:
        pend[0].result(); pf.shutdown()
    ls = [float(x) for x in losses]
    chk = float(sum(p.detach().double().abs().sum() for p in net.parameters()))
    res.append((ls, chk))
    first_diff = next((k for k in range(steps) if ls[k] != res[0][0][k]), None)
    if runs <= 8 or first_diff is not None:
        print("run %d: final loss %.6f checksum %.10e  first step whose loss differs from run 0: %s" % (r, ls[-1], chk, first_diff), flush=True)
import collections
modes = collections.Counter(x[1] for x in res)
print("all runs identical:", all(x == res[0] for x in res), " distinct outcomes:", len(modes), " runs off the most common one: %d of %d" % (runs - modes.most_common(1)[0][1], runs))
