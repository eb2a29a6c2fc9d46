"""How much of the forward's host time is Python / torch glue?  Runs the forward with the two hot
extension entry points replaced by stubs that return cached outputs (no native launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import _ext
from doda_amd.model import SparseConvNet, default_cfg, voxelize_and_run, cross_entropy
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
cfg = default_cfg()
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in make_batch(4, 20000, 1000).items()}
torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev).train()
ext = _ext.ext
def fwd():
    with torch.no_grad():
        return voxelize_and_run(cfg, net, bd, dev, feature_dtype=torch.bfloat16, inputs_ready=True)
def timeit(tag, n=20):
    for _ in range(5): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fwd()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("%-40s %.2f ms host per forward" % (tag, (t1 - t0) / n * 1e3), flush=True)
timeit("real forward (no_grad)")
real_conv, real_bn = ext.indice_conv, ext.bn_relu
cache = {}
def stub_conv(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_fwd, pk_bwd, residual):
    key = ("c", n_out, weight.shape[-1])
    if key not in cache: cache[key] = torch.zeros(n_out, weight.shape[-1], dtype=features.dtype, device=features.device)
    return cache[key]
def stub_bn(x, *a):
    return x
import doda_amd.spconv.functional as F, doda_amd.nn as N
class Stub:  # same attribute surface as the extension module
    def __getattr__(self, k): return getattr(ext, k)
s = Stub(); s.indice_conv = stub_conv; s.bn_relu = stub_bn
F._ext = s; N._ext = s
timeit("conv + BN natives stubbed out")
F._ext = ext; N._ext = ext
timeit("real forward again")
F._ext = s; N._ext = s
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(20): fwd()
pr.disable()
so = io.StringIO(); pstats.Stats(pr, stream=so).sort_stats("cumtime").print_stats(45); print(so.getvalue()[:7000])
