"""Per-process step-time modes: the bench step on a re-used pyramid (no rulebook build) and with the prefetch, with the
host's issue time per step (no synchronisation inside) next to the wall time, and the CPU the issuing thread ran on.
Run several processes back to back: tools/modeprobe.py [steps] [pin=1|0]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.host import pin_to_device_numa
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pin = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
note = pin_to_device_numa(0) if pin else "none"
d = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
wp, wt = True, tile_levels_for(torch.bfloat16)


def run(mode):
    pf = PyramidPrefetcher(d, 7, gated=False)
    pend = [pf.submit(bd, wp, wt, resident=True, now=True)]
    fixed = PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d) if mode == "reuse" else None
    cpus = set()

    def step():
        opt.zero_grad(set_to_none=True)
        if mode == "reuse":
            pyr = fixed
        else:
            pend[0].result()
            pyr = PyramidPrefetcher.take(pend[0], d)
            pend[0] = pf.submit(bd, wp, wt, resident=True)
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
        loss.backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    # (a) free-running: wall per step
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    # (b) host issue time alone: each step issued against an idle GPU (synchronise between steps, outside the clock)
    host = 0.0
    gpu = 0.0
    for _ in range(30):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        a = time.perf_counter()
        e0.record()
        step()
        e1.record()
        host += time.perf_counter() - a
        cpus.add(int(open('/proc/self/stat').read().rsplit(')',1)[1].split()[36]))
        torch.cuda.synchronize()
        gpu += e0.elapsed_time(e1)
    pend[0].result()
    pf.shutdown()
    return wall, host / 30 * 1e3, gpu / 30, sorted(cpus)


for mode in ("reuse", "prefetch", "reuse", "prefetch"):
    w, h, g, c = run(mode)
    print("%-8s wall %.2f ms | issued against an idle GPU: host %.2f ms, GPU first-to-last %.2f ms | cpus %s | pin %s" % (mode, w, h, g, c, note), flush=True)
