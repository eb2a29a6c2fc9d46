import sys; sys.path.insert(0, "/root/repo")
import torch
from doda_amd._lib import lib
from doda_amd import model as M
from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
from doda_amd.scene import make_batch
d = torch.device("cuda:0")
bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(2, 30000, 7).items()}
cfg = default_cfg()
def run(on):
    lib().doda_spconv_set_stats_finish(1 if on else 0)
    torch.manual_seed(0)
    net = SparseConvNet(cfg).to(d).train()
    loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16), bd["labels"])
    loss.backward(); torch.cuda.synchronize()
    return loss.item(), [p.grad.float().cpu() for p in net.parameters()]
def dist(a, b):
    num = sum(float(((x - y) ** 2).sum()) for x, y in zip(a[1], b[1])) ** 0.5
    den = sum(float((y ** 2).sum()) for y in b[1]) ** 0.5
    return num / den
for skip in (True, False):
    M.SKIP_VIA_BN = M.SKIP_IN_BLOCK = skip
    for cat in (True, False):
        M.CAT_STATS = cat
        r = [run(True), run(False), run(True), run(False)]
        print("skip=%d cat=%d: on/off %.4f  on/on %.4f  off/off %.4f  loss %.5f %.5f" % (skip, cat, dist(r[0], r[1]), dist(r[0], r[2]), dist(r[1], r[3]), r[0][0], r[1][0]), flush=True)
