import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
d = torch.device("cuda:0")
torch.manual_seed(0)
p0 = torch.randn(100000, device=d); g1 = torch.randn(100000, device=d); g2 = torch.randn(100000, device=d)
lr, mom, damp, wd = 0.05, 0.9, 0.0, 1e-4
p = torch.nn.Parameter(p0.clone()); o = torch.optim.SGD([p], lr=lr, momentum=mom, dampening=damp, weight_decay=wd, fused=True)
p.grad = g1.clone(); o.step(); t1 = p.detach().clone(); b1 = o.state[p]["momentum_buffer"].clone()
p.grad = g2.clone(); o.step(); t2 = p.detach().clone(); b2 = o.state[p]["momentum_buffer"].clone()
P0, G1, G2 = p0.cpu().numpy(), g1.cpu().numpy(), g2.cpu().numpy()
f32, f64 = np.float32, np.float64
def dbl(P, G, B, first):
    g = (G.astype(f64) + wd * P.astype(f64)).astype(f32)
    b = g.astype(f64) if first else mom * B.astype(f64) + (1 - damp) * g.astype(f64)
    return (P.astype(f64) - lr * b).astype(f32), b.astype(f32)
def dbl_round(P, G, B, first):   # buffer rounded before use
    g = (G.astype(f64) + wd * P.astype(f64)).astype(f32)
    b = (g.astype(f64) if first else mom * B.astype(f64) + (1 - damp) * g.astype(f64)).astype(f32)
    return (P.astype(f64) - lr * b.astype(f64)).astype(f32), b
def flt(P, G, B, first):
    g = G + f32(wd) * P
    b = g if first else f32(mom) * B + f32(1 - damp) * g
    return P - f32(lr) * b, b
for name, fn in (("double", dbl), ("double, rounded buffer", dbl_round), ("float", flt)):
    a1, c1 = fn(P0, G1, None, True); a2, c2 = fn(a1, G2, c1, False)
    print(name, "step1 p mismatches", int((a1 != t1.cpu().numpy()).sum()), "buf", int((c1 != b1.cpu().numpy()).sum()),
          "| step2 p", int((a2 != t2.cpu().numpy()).sum()), "buf", int((c2 != b2.cpu().numpy()).sum()))
from doda_amd.optim import FusedSGD
q = torch.nn.Parameter(p0.clone()); o2 = FusedSGD([q], lr=lr, momentum=mom, dampening=damp, weight_decay=wd)
q.grad = g1.clone(); o2.step(); print("mine step1 p", int((q.detach() != t1).sum()), "buf", int((o2.state[q]["momentum_buffer"] != b1).sum()))
q.grad = g2.clone(); o2.step(); print("mine step2 p", int((q.detach() != t2).sum()), "buf", int((o2.state[q]["momentum_buffer"] != b2).sum()))
