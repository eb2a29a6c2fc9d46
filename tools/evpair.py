import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
for ns in (4, 1):
    b = make_batch(ns, 150000, 1000)
    idx = b["voxel_locs"].int().to(dev); shape = [int(s) for s in b["spatial_shape"]]
    sub = spconv.ops.build_subm(idx, ns, shape, 3); m = idx.shape[0]
    w = torch.randn(27, 16, 16, device=dev) * 0.05
    plan = ops.PackPlan([(w, 27, 16, 16, 0, 2)], dev); plan.run()
    nset = 6 if ns == 4 else 16
    sets = []
    for j in range(nset):
        tbl = sub.tbl.clone()
        sets.append((torch.randn(m, 16, device=dev).bfloat16(), torch.randn(m, 16, device=dev).bfloat16(), tbl, ops.tilebook_build(tbl)))
    def fn(k):
        x, res, tbl, tb = sets[k % nset]
        ops.spconv_gather(x, None, tbl, m, 0, 16, packed=plan.outputs[0], tilebook=tb)
    for k in range(10): fn(k)
    n = 200
    # interval
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for k in range(n): fn(k)
    ev[1].record(); torch.cuda.synchronize()
    interval = ev[0].elapsed_time(ev[1]) * 1e3 / n
    # pairs
    es = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; ee = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda.synchronize()
    for k in range(n):
        es[k].record(); fn(k); ee[k].record()
    torch.cuda.synchronize()
    d = np.array([es[k].elapsed_time(ee[k]) * 1e3 for k in range(n)])
    # pairs with a busy host: sleep between launches
    torch.cuda.synchronize()
    for k in range(n):
        es[k].record(); fn(k); ee[k].record(); time.sleep(0.0002)
    torch.cuda.synchronize()
    d2 = np.array([es[k].elapsed_time(ee[k]) * 1e3 for k in range(n)])
    print("M %d: interval %.2f us | event pairs median %.2f mean %.2f p10 %.2f p90 %.2f | with 200 us host gaps: median %.2f" % (
        m, interval, np.median(d), d.mean(), np.percentile(d, 10), np.percentile(d, 90), np.median(d2)), flush=True)
