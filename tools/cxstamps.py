"""Where one executor op spends its time: wall-clock stamps from inside the kernel (workgroup 0, thread 0)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from doda_amd import ops
from doda_amd._lib import lib
from tests.test_gpu_coarse import _level, _pack, _bf

d = torch.device("cuda:0")
n, c = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (83, 112)
idx, shape, batch = _level(n, n)
n = idx.shape[0]
tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
g = torch.Generator().manual_seed(1)
G = ops.coarse_workgroups()
x = _bf(torch.randn(n, c, generator=g)).to(d)
y = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
w = (torch.randn(27, c, c, generator=g) * 0.05).to(d)
wp = _pack(w, 27, c, c, 0, d)
st = torch.zeros((G, 2, c), dtype=torch.float32, device=d)
B = ops.CX_F_BARRIER
DBG = int(sys.argv[3]) if len(sys.argv) > 3 else 0
gemm = dict(kind=ops.CX_GEMM, flags=B | DBG, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, x=x, w=wp, tbl=tbl, y=y, stats=st)
buf = torch.zeros(4001, dtype=torch.int64, device=d)
ops.coarse_run([gemm] * 4, d)
torch.cuda.synchronize()
lib().doda_coarse_debug_stamps(buf.data_ptr())
ops.coarse_run([gemm] * 4, d)
torch.cuda.synchronize()
lib().doda_coarse_debug_stamps(None)
h = buf.cpu().numpy().astype(np.uint64)
k = int(h[0])
NAMES = {1: "op: before barrier", 2: "op: after barrier", 3: "op: done", 20: "gemm entered", 21: "decode table ready", 10: "unit start", 11: "strip in LDS", 12: "rows requested", 13: "chunk parked",
         14: "chunk visible", 15: "chunk multiplied", 16: "chunk loop done", 17: "epilogue done"}
t0 = None
prev = None
for i in range(k):
    v = int(h[1 + i])
    lab, t = v >> 56, v & ((1 << 56) - 1)
    if t0 is None:
        t0 = prev = t
    print("%8.2f us  (+%6.2f)  %s" % ((t - t0) / 100.0, (t - prev) / 100.0, NAMES.get(lab, lab)))
    prev = t
