#!/usr/bin/env python
"""Where the zero-change route's time goes (bench.py `reference_graph`): the reference's module tree (doda_amd.refgraph) with the
glue around it switched, one piece at a time, from the reference's (torch CrossEntropyLoss, torch.optim.SGD, immediate weight
gradients, rulebooks inside the convs) to doda_amd's (fused loss, one-launch SGD, deferred weight gradients, prefetched rulebooks),
and doda_amd.model.SparseConvNet at the end.  usage: python tools/refgraph_ab.py [--dtype bf16|f32] [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
from doda_amd import spconv
from doda_amd.host import pin_to_device_numa
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.refgraph import RefSparseConvNet, run_reference_route
from doda_amd.scene import make_batch
from doda_amd.spconv import functional as Fsp

pin_to_device_numa(0)
dev = torch.device("cuda:0")
fdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
cfg = default_cfg()
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}


def timed(step, warm=6):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3


def ref_variant(fused_loss, fused_sgd, deferred, prebuilt):
    torch.manual_seed(0)
    net = RefSparseConvNet(cfg).to(dev).train()
    opt = (FusedSGD if fused_sgd else torch.optim.SGD)(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    Fsp.set_deferred_wgrad(deferred)
    pre = PyramidPrefetcher(dev, 7) if prebuilt else None
    pend = [pre.submit(bd, False, tile_levels_for(fdt), resident=True, now=True)] if pre else None

    def step():
        opt.zero_grad(set_to_none=True)
        if pre:
            idx32, book = PyramidPrefetcher.take(pend[0], dev)
            pend[0] = pre.submit(bd, False, tile_levels_for(fdt), resident=True)
            from doda_amd import pointgroup_ops
            vf = pointgroup_ops.voxelization(bd["feats"], bd["v2p_map"], 4)
            inp = spconv.SparseConvTensor(vf.to(fdt), idx32, bd["spatial_shape"], 4)
            inp.indice_dict.update(book)
            scores = net(inp, bd["p2v_map"])
        else:
            scores = run_reference_route(cfg, net, bd, dev, feature_dtype=fdt)
        loss = cross_entropy(scores.float(), bd["labels"]) if fused_loss else crit(scores.float(), bd["labels"])
        loss.backward()
        opt.step()
    try:
        return timed(step)
    finally:
        Fsp.set_deferred_wgrad(False)
        if pre:
            pend[0].result()
            pre.shutdown()


print("reference tree, reference glue                         : %.2f ms" % ref_variant(False, False, False, False), flush=True)
print("  + fused cross-entropy                                : %.2f ms" % ref_variant(True, False, False, False), flush=True)
print("  + one-launch SGD                                     : %.2f ms" % ref_variant(True, True, False, False), flush=True)
print("  + deferred weight gradients                          : %.2f ms" % ref_variant(True, True, True, False), flush=True)
print("  + rulebooks prefetched on the side stream            : %.2f ms" % ref_variant(True, True, True, True), flush=True)
