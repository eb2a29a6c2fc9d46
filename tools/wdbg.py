import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_round3 import _raster_scene
from doda_amd import ops
d = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
shape, batch = [80, 70, 60], 2
idx = torch.from_numpy(_raster_scene(5 + m, m, batch, shape)).to(d)
tbl = ops.rulebook_subm(idx, shape, batch, 3)
n = tbl.shape[1]
torch.manual_seed(0)
tb = ops.tilebook_build(tbl)
t = tbl.cpu().long()
for trial in range(3):
    x = torch.randn(n, 16, device=d).bfloat16(); dy = torch.randn(n, 16, device=d).bfloat16()
    x2 = torch.randn(n, 16, device=d).bfloat16()
    outs = ops.spconv_wgrad_multi([(x, dy, tbl, n, None, None, tb), (x2, dy, tbl, n, None, None, tb)])
    torch.cuda.synchronize()
    for k, xx in enumerate((x, x2)):
        ref = torch.zeros(27, 16, 16, dtype=torch.float64)
        for o in range(27):
            sel = t[o] >= 0
            ref[o] = xx.double().cpu()[t[o][sel]].t() @ dy.double().cpu()[sel]
        got = outs[k].cpu().double().reshape(27, 16, 16)
        bad = ~torch.isfinite(got) | ((got - ref).abs() > 1e-3 * ref.abs().max())
        print("trial", trial, "job", k, "n", n, "bad entries", int(bad.sum()), "nan", int((~torch.isfinite(got)).sum()),
              "offsets with bad:", sorted(set(bad.nonzero()[:, 0].tolist()))[:30], "ci:", sorted(set(bad.nonzero()[:, 1].tolist())), "co:", sorted(set(bad.nonzero()[:, 2].tolist())))
