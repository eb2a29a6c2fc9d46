"""What the issuing thread does per step besides launching: caching-allocator calls, autograd nodes, extension calls — and what
one of each costs on this host (torch.empty, a raw launch through the C ABI, an empty pybind call).  tools/hostcount.py [voxels]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import model as M, spconv
from doda_amd.host import pin_to_device_numa
from doda_amd.optim import FusedSGD
from doda_amd.scene import make_batch
pin_to_device_numa(0)
vox = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = torch.device("cuda:0")
cfg = M.default_cfg(); torch.manual_seed(0)
net = M.SparseConvNet(cfg).to(d).train()
opt = FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
spconv.functional.set_deferred_wgrad(True)
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(4, vox, 1000).items()}
pf = M.PyramidPrefetcher(d, 7)
wp, wt = True, M.tile_levels_for(torch.bfloat16)
pyr = M.PyramidPrefetcher.take(pf.submit(bd, wp, wt, resident=True, now=True), d)


def step():
    opt.zero_grad(set_to_none=True)
    a = time.perf_counter()
    loss = M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step()
    return loss, b - a, c - b, time.perf_counter() - c


for _ in range(10): step()
torch.cuda.synchronize()
from doda_amd._ext import ext as _e
if hasattr(_e, "host_timing"): _e.host_timing()        # (reset; DODA_HOST_TIMING=1 fills it)
s0 = torch.cuda.memory_stats(d)["allocation.all.allocated"]
n = 20; tf = tb = to = 0.0
for _ in range(n):
    l, f, b, o = step(); tf += f; tb += b; to += o
torch.cuda.synchronize()
s1 = torch.cuda.memory_stats(d)["allocation.all.allocated"]
if hasattr(_e, "host_timing"):
    ht = _e.host_timing()
    if any(v[1] for v in ht.values()):
        print("extension entry points, host us per step: " + "; ".join("%s %.0f (%d calls)" % (k, v[0] / n, v[1] / n) for k, v in ht.items()))
print("host per step (re-used pyramid): fwd %.2f ms, bwd %.2f ms, opt %.2f ms; allocator calls per step: %.0f" % (tf / n * 1e3, tb / n * 1e3, to / n * 1e3, (s1 - s0) / n))
# autograd nodes of one step
loss = M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16, inputs_ready=True, pyramid=pyr), bd["labels"])
seen, stack, names = set(), [loss.grad_fn], {}
while stack:
    f = stack.pop()
    if f is None or f in seen: continue
    seen.add(f); names[f.name()] = names.get(f.name(), 0) + 1
    stack.extend(g for g, _ in f.next_functions)
print("autograd nodes per step: %d  %s" % (len(seen), sorted(names.items(), key=lambda kv: -kv[1])[:12]))
# unit costs
t0 = time.perf_counter()
for _ in range(2000): x = torch.empty(1024, device=d)
t1 = time.perf_counter()
from doda_amd._lib import lib
from doda_amd import ops
g = torch.rand(16, device=d); xx = torch.randn(4096 * 4, 16, device=d).bfloat16(); yy = torch.empty_like(xx)
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(2000): lib().doda_bn_relu_apply(xx.data_ptr(), xx.shape[0], 16, 2, g.data_ptr(), g.data_ptr(), g.data_ptr(), g.data_ptr(), 1, yy.data_ptr(), st)
t3 = time.perf_counter(); torch.cuda.synchronize()
from doda_amd._ext import ext
t4 = time.perf_counter()
for _ in range(2000): ext.get_defer_wgrad()
t5 = time.perf_counter()
print("torch.empty %.2f us; C-ABI launch through ctypes %.2f us; empty pybind call %.2f us" % ((t1 - t0) / 2000 * 1e6, (t3 - t2) / 2000 * 1e6, (t5 - t4) / 2000 * 1e6))
pf.shutdown()
