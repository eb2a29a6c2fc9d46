import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from doda_amd.train import DeviceMeters
d = torch.device("cuda:0")
n, k = 781042, 20
g = torch.Generator().manual_seed(0)
labels = torch.randint(0, k, (n,), generator=g).to(d); labels[::17] = 255
preds = torch.randint(0, k, (n,), generator=g).to(d)
loss = torch.tensor(1.0, device=d)
m = DeviceMeters(k, 255, d)  # (one launch since ABI 12; 612 us with three scatter_add_)
for _ in range(5): m.update(loss, preds, labels)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): m.update(loss, preds, labels)
e1.record(); torch.cuda.synchronize()
print("DeviceMeters.update: %.1f us per call (GPU)" % (e0.elapsed_time(e1) / 50 * 1e3))
t = time.perf_counter()
for _ in range(50): m.update(loss, preds, labels)
print("host issue: %.1f us per call" % ((time.perf_counter() - t) / 50 * 1e6)); torch.cuda.synchronize()
