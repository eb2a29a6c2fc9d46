import sys, collections
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_gpu_unet import _run
g, scores, loss, grads, net = _run(torch.float32)
lv = collections.defaultdict(float); worst = {}
for name, ref in zip(g["grad_names"], g["grad_norms"]):
    name = str(name)
    depth = name.split(".").count("u")
    kind = "bn" if (name.endswith(".bias") or (name.endswith(".weight") and grads[name] is not None and net.state_dict()[name].dim() == 1)) else "w"
    e = abs(grads[name] - ref) / (ref + 1e-12)
    if e > lv[(depth, kind)]:
        lv[(depth, kind)] = e; worst[(depth, kind)] = name
for k in sorted(lv): print(k, "%.3e" % lv[k], worst[k])
