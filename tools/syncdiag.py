"""Diagnosis: how much of a step is lost to the rulebook size read-backs (host blocks behind the
previous step's queue)?  Compares the normal step with one whose rulebook pyramid is cached."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from doda_amd import spconv
from doda_amd.model import SparseConvNet, default_cfg, voxelize_and_run, cross_entropy
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
cfg = default_cfg()
batch = make_batch(4, 150000, 1000)
batch_dev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
torch.manual_seed(0)
net = SparseConvNet(cfg).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
labels = batch_dev["labels"]
def step():
    opt.zero_grad(set_to_none=True)
    scores = voxelize_and_run(cfg, net, batch_dev, dev, feature_dtype=torch.bfloat16,
                              inputs_ready=os.environ.get("SIDE", "1") == "1")
    loss = cross_entropy(scores, labels, ignore_index=255)
    loss.backward(); opt.step()
    return loss
def run(tag, n=30):
    for _ in range(8): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); print("%-28s %.2f ms/step" % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)
for rep in range(3):
    for side in ("0", "1"):
        os.environ["SIDE"] = side
        run("side-stream pyramid = %s" % side, 40)
