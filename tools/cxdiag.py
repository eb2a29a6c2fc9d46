"""Executor vs per-layer path: per-parameter gradient distance of one U-Net step (tools, GPU)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_coarse import _unet_step

level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
vox = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
_, lf, gf, _ = _unet_step(False, level, dtype=torch.float32, voxels=vox)
_, l0, g0, _ = _unet_step(False, level, voxels=vox)
_, l1, g1, _ = _unet_step(True, level, voxels=vox)
print("loss fp32 %.6f  bf16 layer %.6f  bf16 exec %.6f" % (lf, l0, l1))
pre = "unet." + "u." * (level - 2)
for n in g0:
    if not n.startswith(pre):
        continue
    r = lambda a, b: float((a - b).norm() / b.norm().clamp(min=1e-20))
    print("%-60s |g| %.3e  exec-layer %.3f  layer-fp32 %.3f  exec-fp32 %.3f" % (n, float(gf[n].norm()), r(g1[n], g0[n]), r(g0[n], gf[n]), r(g1[n], gf[n])))
