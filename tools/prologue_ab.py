"""In-process A/B of the BatchNorm prologue (doda_conv_epilogue.pre_*, DODA_BN_PROLOGUE): the bench's bf16 training step
(bench.py run_training: prefetched rulebooks, deferred weight gradients, FusedSGD) with the BatchNorm apply pass inside
the tile kernels' staging against the separate apply launches, alternating inside ONE process on ONE box.
  python tools/prologue_ab.py [steps] [rounds]      -> ms/step per configuration and round"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd.host import pin_to_device_numa
pin_to_device_numa(0)
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
from doda_amd.spconv import functional as Fsp
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
which = sys.argv[3] if len(sys.argv) > 3 else "prologue"   # prologue | skip (model.SKIP_VIA_BN / SKIP_IN_BLOCK)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
Fsp.set_deferred_wgrad(True)
PF = PyramidPrefetcher(dev, 7)
tiles = tile_levels_for(torch.bfloat16)
pend = [PF.submit(bd, True, tiles, resident=True, now=True)]


def step():
    opt.zero_grad(set_to_none=True)
    pyr = PyramidPrefetcher.take(pend[0], dev); pend[0] = PF.submit(bd, True, tiles, resident=True)
    l = cross_entropy(voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
    l.backward(); opt.step()


def run(n, on):
    if which == "skip":
        import doda_amd.model as M
        M.SKIP_VIA_BN = M.SKIP_IN_BLOCK = on
    else:
        Fsp.set_bn_prologue(on)
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(20): step()
for rnd in range(rounds):
    print("round %d: " % rnd + "  ".join("%s=%d %.3f ms" % (which, on, run(steps, bool(on))) for on in (1, 0, 1, 0)), flush=True)
pend[0].result()
PF.shutdown()
