"""Shared helpers for the tests: seeded synthetic voxel sets."""
import numpy as np


def random_voxels(seed, n, batch, shape):
    """n unique (b,x,y,z) int32 rows in random order."""
    rng = np.random.default_rng(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    n = min(n, cells)
    lin = rng.choice(cells, size=n, replace=False)
    z = lin % shape[2]
    y = (lin // shape[2]) % shape[1]
    x = (lin // (shape[2] * shape[1])) % shape[0]
    b = lin // (shape[2] * shape[1] * shape[0])
    return np.stack([b, x, y, z], 1).astype(np.int32)


def surface_voxels(seed, n, batch, shape):
    """Voxels concentrated on a few planes (ScanNet-like neighbourhood statistics)."""
    rng = np.random.default_rng(seed)
    pts = set()
    while len(pts) < n:
        b = int(rng.integers(batch))
        axis = int(rng.integers(3))
        level = int(rng.integers(shape[axis]))
        for _ in range(64):
            p = [int(rng.integers(shape[0])), int(rng.integers(shape[1])), int(rng.integers(shape[2]))]
            p[axis] = min(shape[axis] - 1, level + int(rng.integers(2)))
            pts.add((b, p[0], p[1], p[2]))
            if len(pts) >= n:
                break
    arr = np.array(list(pts), dtype=np.int32)
    rng.shuffle(arr)
    return arr


def deterministic_init(model, seed=0):
    """Re-initialise every parameter / BN buffer from one CPU generator, visiting state-dict keys in
    sorted order, so two structurally equivalent models (the reference's SparseConvNet and
    doda_amd.model.SparseConvNet) get identical weights regardless of construction order."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if key.endswith("num_batches_tracked"):
                t.zero_()
            elif key.endswith("running_mean"):
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif key.endswith("running_var"):
                t.copy_(1.0 + 0.2 * torch.rand(t.shape, generator=g))
            elif t.dim() == 1:  # BN affine / biases
                base = 1.0 if key.endswith("weight") else 0.0
                t.copy_(base + 0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t.shape[-2] * (t.numel() // (t.shape[-1] * t.shape[-2])) if t.dim() == 5 else t.shape[1]
                t.copy_(torch.randn(t.shape, generator=g) * (1.5 / max(fan_in, 1)) ** 0.5)
    return model


def stats_sums(st):
    """[2, c] float64 sums of a conv epilogue's BatchNorm statistics in either form the extension hands out: rows
    [n, 2, c] fp32 (one per workgroup) or — ABI 9, the default — fp64 totals [8, 2, c / 4, 16] (doda_amd.ops.totals_sums)."""
    if st.dim() == 4:
        return st[..., :4].sum(0).reshape(2, -1)
    return st.double().sum(0)
