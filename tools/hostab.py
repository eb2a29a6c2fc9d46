"""In-process A/B of a host-side switch on a small scene (the step is then paced by the issuing thread alone):
alternating blocks of steps with a switch (model.FAST_BLOCKS) on / off; prints the host issue time per step of each block.
tools/hostab.py [voxels] [steps] [rounds] [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import model as M, spconv
from doda_amd.host import pin_to_device_numa
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
pin_to_device_numa(0)
vox = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
SWITCH = sys.argv[4] if len(sys.argv) > 4 else "blocks"     # blocks: model.FAST_BLOCKS
from doda_amd._lib import lib
d = torch.device("cuda:0")
cfg = M.default_cfg(); torch.manual_seed(0)
net = M.SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, vox, 1000).items()}
pf = M.PyramidPrefetcher(d, 7)
wp, wt = True, M.tile_levels_for(torch.bfloat16)
pend = [pf.submit(bd, wp, wt, resident=True, now=True)]


def step():
    a = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    pyr = M.PyramidPrefetcher.take(pend[0], d)
    pend[0] = pf.submit(bd, wp, wt, resident=True)
    loss = M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step()
    return b - a, c - b, time.perf_counter() - c


for _ in range(10):
    step()
res = {True: [], False: []}
for r in range(rounds):
    for fast in (True, False):
        if SWITCH == "blocks":
            M.FAST_BLOCKS = fast
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t = [0.0, 0.0, 0.0]
        t0 = time.perf_counter()
        for _ in range(steps):
            f, b, o = step()
            t[0] += f; t[1] += b; t[2] += o
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        res[fast].append((t[0] / steps * 1e3, t[1] / steps * 1e3, t[2] / steps * 1e3, wall))
for fast in (True, False):
    print("%s=%d" % (SWITCH, fast), " | ".join("fwd %.2f bwd %.2f opt %.2f wall %.2f" % x for x in res[fast]))
pend[0].result(); pf.shutdown()
