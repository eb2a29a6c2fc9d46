import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.util import deterministic_init
from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch
from oracle.unet_cpu import OracleUNet
from tests.test_gpu_round2 import forward_backward_loss
d = torch.device('cuda:0')
batch = make_batch(2, 3000, 77)
cfg = default_cfg()
net = deterministic_init(SparseConvNet(cfg), seed=0).to(d).train()
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items()}
loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.float32), bd["labels"]); loss.backward()
ref = deterministic_init(OracleUNet(), seed=0).double().train()
import oracle.spconv_cpu as sp
from oracle import oracle as orc
vf = torch.from_numpy(orc.voxelize_fp(batch["feats"].numpy(), batch["v2p_map"].numpy(), True)).double()
inp = sp.SparseConvTensor(vf, batch["voxel_locs"].int(), batch["spatial_shape"], 2)
l2 = torch.nn.functional.cross_entropy(ref(inp, batch["p2v_map"]), batch["labels"], ignore_index=255); l2.backward()
print("loss", float(loss), float(l2))
rp = dict(ref.named_parameters())
worst = []
for k, p in net.named_parameters():
    a = p.grad.double().cpu(); b = rp[k].grad
    e = float((a-b).abs().max() / b.abs().max().clamp_min(1e-30))
    worst.append((e, k, float(b.abs().max())))
worst.sort(reverse=True)
for e,k,m in worst[:12]: print("%.3e %-50s max|g| %.3e" % (e,k,m))
