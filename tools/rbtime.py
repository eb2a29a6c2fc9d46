import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from doda_amd import spconv
from doda_amd.scene import make_batch
dev = torch.device("cuda:0")
batch = make_batch(4, 150000, 1000)
idx = batch["voxel_locs"].int().to(dev); shape = [int(s) for s in batch["spatial_shape"]]
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
print("subm L1 %.1f us" % t(lambda: spconv.ops.build_subm(idx, 4, shape, 3)))
print("down2 L1 %.1f us" % t(lambda: spconv.ops.build_down2(idx, 4, shape, 2, 2, 0, 1)))
from doda_amd.spconv.core import SparseConvTensor
def pyr():
    x = SparseConvTensor(None, idx, shape, 4); spconv.ops.build_pyramid(x, 7)
print("pyramid %.1f us" % t(pyr))
