"""UBlock(level) subtree: executor vs per-layer bf16 vs per-layer fp32 (ground truth) — output, input gradient, parameter gradients."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_coarse import _subtree, _run_subtree, _bf, dev

for level, n in ((7, 83), (6, 420), (5, 1900), (4, 8400)):
    net, ub, ind, shape, batch = _subtree(level, n, 17)
    g = torch.Generator().manual_seed(level * 1000 + n)
    c = 16 * level
    x0 = _bf(torch.randn(ind.shape[0], c, generator=g)).to(dev())
    gout = _bf(torch.randn(ind.shape[0], c, generator=g)).to(dev())
    state = {k: v.clone() for k, v in ub.state_dict().items()}
    yf, dxf, gf, _ = _run_subtree(ub, ind, shape, batch, level, x0.float(), gout.float(), False)
    ub.load_state_dict(state)
    y0, dx0, g0, _ = _run_subtree(ub, ind, shape, batch, level, x0, gout, False)
    ub.load_state_dict(state)
    y1, dx1, g1, _ = _run_subtree(ub, ind, shape, batch, level, x0, gout, True)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp(min=1e-20))
    print("level %d rows %d: y exec-layer %.4f layer-fp32 %.4f exec-fp32 %.4f | dx exec-layer %.4f layer-fp32 %.4f exec-fp32 %.4f" % (
        level, ind.shape[0], rel(y1, y0), rel(y0, yf), rel(y1, yf), rel(dx1, dx0), rel(dx0, dxf), rel(dx1, dxf)))
    for k in g0:
        print("   %-52s exec-layer %.4f  layer-fp32 %.4f  exec-fp32 %.4f" % (k, rel(g1[k], g0[k]), rel(g0[k], gf[k]), rel(g1[k], gf[k])))
