"""BN(+ReLU) fwd/bwd at the U-Net's level sizes (rows x channels), for tools/ktrace-style profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from doda_amd import nn as dnn
dev = torch.device("cuda:0")
sizes = [(602508, 16), (602508, 32), (183434, 32), (183434, 64), (46084, 48), (11221, 64), (2537, 80)]
for m, c in sizes:
    bn = torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1).to(dev)
    x = torch.randn(m, c, device=dev).bfloat16().requires_grad_(True)
    g = torch.randn(m, c, device=dev).bfloat16()
    for _ in range(8):
        y = dnn.batch_norm_relu(x, bn, True)
        y.backward(g)
torch.cuda.synchronize()
print("done")
