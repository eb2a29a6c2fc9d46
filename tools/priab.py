"""In-process A/B: the bench step issued on the default stream vs on a HIGH-priority stream (the rulebook prefetch stays on
a normal-priority side stream), alternating blocks.  tools/priab.py [steps] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import model as M, spconv
import doda_amd.streams as S
from doda_amd.host import pin_to_device_numa
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
pin_to_device_numa(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.device("cuda:0")
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
cfg = M.default_cfg(); torch.manual_seed(0)
net = M.SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, 150000, 1000).items()}
wp, wt = True, M.tile_levels_for(torch.bfloat16)
torch.cuda.synchronize()


def run(main_stream, side_priority=None, reuse=False):
    with torch.cuda.stream(main_stream):
        S._CACHE.clear()
        pf = M.PyramidPrefetcher(d, 7)
        if side_priority is not None:
            pf.stream = torch.cuda.Stream(device=d, priority=side_priority)
        pend = [pf.submit(bd, wp, wt, resident=True, now=True)]
        fixed = M.PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d) if reuse else None

        def step():
            opt.zero_grad(set_to_none=True)
            if reuse:
                pyr = fixed
            else:
                pyr = M.PyramidPrefetcher.take(pend[0], d)
                pend[0] = pf.submit(bd, wp, wt, resident=True)
            loss = M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
            loss.backward()
            opt.step()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        pend[0].result(); pf.shutdown()
    torch.cuda.synchronize()
    return dt


lo, hi = torch.cuda.Stream.priority_range()
hp = torch.cuda.Stream(device=d, priority=hi)
np_ = torch.cuda.Stream(device=d)
res = {}
for r in range(rounds):
    for name, fn in (("default-main", lambda: run(torch.cuda.default_stream(d))),
                     ("hp-main", lambda: run(hp)),
                     ("hp-main+lp-side", lambda: run(hp, side_priority=lo)),
                     ("default-main+lp-side", lambda: run(torch.cuda.default_stream(d), side_priority=lo)),
                     ("plain-main", lambda: run(np_)),
                     ("reuse(default)", lambda: run(torch.cuda.default_stream(d), reuse=True)),
                     ("reuse(hp)", lambda: run(hp, reuse=True))):
        res.setdefault(name, []).append(fn())
for k, v in res.items():
    print("%-22s" % k, " ".join("%.2f" % x for x in v), "| median %.2f" % sorted(v)[len(v) // 2])
