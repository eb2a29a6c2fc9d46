#!/usr/bin/env python
"""A/B of the coarse-level backends on the bench step (DODA_COARSE_MODE: off / layers / exec), in-process, alternating blocks.
usage: python tools/layers_ab.py [--dtype bf16|f32] [--scenes 4] [--steps 40] [--modes off,layers] [--level 4]
Prints ms per step per block and the per-layer backend's launch counts (forward, backward) of the last step."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--scenes", type=int, default=4)
ap.add_argument("--voxels", type=int, default=150000)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--modes", default="off,layers")
ap.add_argument("--level", type=int, default=4)
ap.add_argument("--heads", default="fused", help="comma list of head forms to alternate: fused (voxel-level head + loss), matrix")
args = ap.parse_args()

from doda_amd import model as M
from doda_amd.host import pin_to_device_numa
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
from doda_amd.spconv import functional as Fsp
from doda_amd._ext import ext

pin_to_device_numa(0)
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
cfg = default_cfg()
batch = make_batch(args.scenes, args.voxels, 1000)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
assert Fsp.set_deferred_wgrad(True)
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
pre = PyramidPrefetcher(dev, len(net.unet.nPlanes))
with_pairs = Fsp.WGRAD_PAIRS and dtype == torch.bfloat16


HEAD = ["fused"]


def step(fut):
    pyr = PyramidPrefetcher.take(fut, dev)
    nxt = pre.submit(bd, with_pairs=with_pairs, with_tiles=tile_levels_for(dtype), resident=True)
    opt.zero_grad(set_to_none=True)
    if HEAD[0] == "fused":
        loss = voxelize_and_run(cfg, net, bd, dev, feature_dtype=dtype, pyramid=pyr, labels=bd["labels"])
    else:
        scores = voxelize_and_run(cfg, net, bd, dev, feature_dtype=dtype, pyramid=pyr)
        loss = cross_entropy(scores, bd["labels"])
    loss.backward()
    opt.step()
    return nxt, loss


fut = pre.submit(bd, with_pairs=with_pairs, with_tiles=tile_levels_for(dtype), resident=True, now=True)
modes = args.modes.split(",")
for m in modes:          # warm-up of every mode
    M.set_coarse_mode(m, args.level)
    for _ in range(10):
        fut, loss = step(fut)
torch.cuda.synchronize()
for r in range(args.rounds):
  for head in args.heads.split(","):
    HEAD[0] = head
    for m in modes:
        M.set_coarse_mode(m, args.level)
        for _ in range(3):
            fut, loss = step(fut)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fut, loss = step(fut)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        print("round %d mode %-7s head %-6s level %d: %.3f ms/step  loss %.5f  launches (fwd, bwd) of the op lists: %s" %
              (r, m, head, args.level, dt, float(loss), ext.coarse_launches() if m == "layers" else "-"), flush=True)
pre.shutdown()
