import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doda_amd import ops, spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
nvox = int(sys.argv[1]); nc = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = make_batch(1, nvox, 1000)
idx = batch["voxel_locs"].int().to(dev); shape = [int(s) for s in batch["spatial_shape"]]
sub = spconv.ops.build_subm(idx, 1, shape, 3)
m = idx.shape[0]
torch.cuda.synchronize(); print("M", m, flush=True)
x = torch.randn(m, 16, device=dev).bfloat16(); w = torch.randn(27, 16, nc, device=dev) * 0.1
y = ops.spconv_gather(x, w, sub.tbl, m, 0, nc); torch.cuda.synchronize(); print("ran", flush=True)
xf = torch.cat([x.float(), torch.zeros(1, 16, device=dev)]); wb = w.bfloat16().float()
ref = torch.zeros(m, nc, device=dev)
for o in range(27):
    t = sub.tbl[o].long(); t = torch.where(t < 0, torch.full_like(t, m), t)
    ref += xf[t] @ wb[o]
print("max err", float((y.float() - ref).abs().max()), "ref max", float(ref.abs().max()), flush=True)
