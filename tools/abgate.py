"""In-process A/B of the rulebook prefetch gate: the bench step with the next pyramid built (a) from the start of the
step, (b) from the step's coarse phase on (PyramidPrefetcher(gated=True)), (c) not at all (pyramid re-used: the floor).
Alternating blocks of `steps` steps, several rounds: box-to-box and run-to-run noise (+-0.5 ms) cancels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import spconv
from doda_amd.model import PyramidPrefetcher, SparseConvNet, cross_entropy, default_cfg, tile_levels_for, voxelize_and_run
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.device("cuda:0")
cfg = default_cfg(); torch.manual_seed(0)
net = SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
wp, wt = True, tile_levels_for(torch.bfloat16)


def run(mode):
    pf = PyramidPrefetcher(d, 7, gated=(mode == "gated")) if mode != "reuse" else PyramidPrefetcher(d, 7, gated=False)
    pend = [pf.submit(bd, wp, wt, resident=True, now=True)]
    fixed = PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d) if mode == "reuse" else None

    acc = [0.0, 0.0, 0.0]

    def step():
        opt.zero_grad(set_to_none=True)
        if mode == "reuse":
            pyr = fixed
        else:
            a = time.perf_counter()
            pend[0].result()
            b = time.perf_counter()
            pyr = PyramidPrefetcher.take(pend[0], d)
            c = time.perf_counter()
            pend[0] = pf.submit(bd, wp, wt, resident=True)
            e = time.perf_counter()
            acc[0] += b - a; acc[1] += c - b; acc[2] += e - c
        loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
        loss.backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    pend[0].result()
    pf.shutdown()
    if mode != "reuse":
        print("   %s: per step wait-for-build %.3f ms, take %.3f ms, submit %.3f ms" % (mode, acc[0] / (steps + 10) * 1e3, acc[1] / (steps + 10) * 1e3, acc[2] / (steps + 10) * 1e3))
    return dt


res = {"start": [], "gated": [], "reuse": []}
for r in range(rounds):
    for mode in ("start", "gated", "reuse"):
        res[mode].append(run(mode))
for k, v in res.items():
    print(k, " ".join("%.2f" % x for x in v), "| median %.2f" % sorted(v)[len(v) // 2])

if os.environ.get("QUEUE_SCAN") == "1":
    keep = []
    for k in range(8):
        import doda_amd.streams as _st
        _st._CACHE.clear()          # a fresh calibration per run
        print("streams created before this run:", len(keep), "-> start %.2f ms" % run("start"), flush=True)
        keep.append(torch.cuda.Stream(device=d))
