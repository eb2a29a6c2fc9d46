"""Phase time stamps of one conv_dma16 workgroup (DODA_DMA_DBG=128): per iteration the shader-clock deltas
wait | barrier | multiply + DMA issue | list/operand issue | epilogue + stores, for waves 0 and 5."""
import ctypes as C, json, os, sys
os.environ["DODA_DMA_DBG"] = str(128 | int(os.environ.get("DBG_EXTRA", "0")))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doda_amd import ops, spconv
from doda_amd._lib import lib
from doda_amd.scene import make_batch
d = torch.device("cuda:0")
b = make_batch(4, 150000, 1000)
idx = b["voxel_locs"].int().to(d)
data = spconv.ops.build_subm(idx, 4, b["spatial_shape"], 3)
m = idx.shape[0]
w = torch.randn(27, 16, 16, device=d) * 0.1
plan = ops.PackPlan([(w, 27, 16, 16, 0, 2)], d); plan.run(); pk = plan.outputs[0]
n = 6
xs = [torch.randn(m, 16, device=d).bfloat16() for _ in range(n)]
ys = [torch.empty(m, 16, device=d, dtype=torch.bfloat16) for _ in range(n)]
tbls = [data.tbl.clone() for _ in range(n)]
tbs = [ops.tilebook_build(t) for t in tbls]
for r in range(12):
    j = r % n
    ops.spconv_gather(xs[j], None, tbls[j], m, 0, 16, packed=pk, tilebook=tbs[j], out=ys[j])
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
assert lib().doda_debug_dma_stamps(buf) == 0
for wv in range(2):
    print("wave", (0, 5)[wv])
    prev_end = None
    for it in range(10):
        s = [buf[(wv * 16 + it) * 8 + p] for p in range(6)]
        if s[0] == 0:
            break
        d_ = [s[k + 1] - s[k] for k in range(5)]
        gap = (s[0] - prev_end) if prev_end else 0
        prev_end = s[5]
        print("  it %2d: gap %5d | wait %6d | barrier %6d | multiply+dma %6d | list/epi %5d | epilogue %6d | total %6d" % (
            it, gap, *d_, s[5] - s[0]))
