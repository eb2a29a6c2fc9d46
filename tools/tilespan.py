import sys; sys.path.insert(0, "/root/repo")
import torch
from doda_amd import spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev)
shape = [int(s) for s in batch["spatial_shape"]]
sub = spconv.ops.build_subm(idx, 4, shape, 3)
t = sub.tbl.long()
n = t.shape[1]; nt = (n + 255) // 256
pad = nt * 256 - n
big = torch.full((27, pad), -1, device=dev, dtype=torch.long)
tt = torch.cat([t, big], 1).view(27, nt, 256)
mx = tt.amax(dim=(0, 2))
mn = torch.where(tt >= 0, tt, torch.full_like(tt, 1 << 40)).amin(dim=(0, 2))
span = (mx - mn + 1).float()
q = torch.quantile(span, torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev))
print("tiles", nt, "span quantiles 10/50/90/99 %:", q.tolist(), "max", span.max().item(), "share <= 131072:", (span <= 131072).float().mean().item())
