"""GPU tests of the BatchNorm(+ReLU) PROLOGUE of the sparse-convolution kernels (ABI 6, SURVEY §8 f1: the
BN -> ReLU -> conv triple of reference model/unet_block.py:23-30,46-49 as ONE conv launch + the BatchNorm's reduction).

* kernel level, through the C ABI (doda_spconv_gather_ex with doda_conv_epilogue.pre_*): the normalised side output is
  BIT-equal to the BatchNorm arithmetic written out in torch fp32 (sub, mul, mul, add, ReLU, one bf16 rounding), the
  conv output BIT-equal to the same kernel fed that tensor, and within the bf16 rounding step of the fp64 oracle of
  BN -> ReLU -> conv; statistics epilogue and residual add unchanged; ragged last tile, 16 / 32 input channels, NB = 2,
  tiles without a list (dense-table fallback inside the kernel);
* doda_bn_fwd_final against the two-pass BatchNorm's save_mean / save_invstd / running statistics;
* extension level: a U-Net training step with the prologue on and off is BIT-equal (loss, gradients, buffers);
* shapes the prologue does not cover are refused (DODA_ERR_UNSUPPORTED), never silently computed another way.
"""
import numpy as np
import pytest
import torch

from tests.util import surface_voxels

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _ext_or_skip():
    from doda_amd._ext import ext
    if ext is None:
        pytest.skip("compiled extension not built")
    return ext


def _scene_table(m, seed=0, batch=2, shape=(80, 70, 60)):
    from doda_amd import ops
    d = dev()
    idx = torch.from_numpy(surface_voxels(seed, m, batch, list(shape))).to(d)
    key = ((idx[:, 0].long() * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]
    idx = idx[torch.argsort(key)].contiguous()
    return idx, ops.rulebook_subm(idx, list(shape), batch, 3)


def _bn_vectors(c, d, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    mean = (torch.randn(c, generator=g) * 0.3).to(d)
    invstd = (torch.rand(c, generator=g) + 0.5).to(d)
    gamma = (torch.rand(c, generator=g) + 0.5).to(d)
    beta = (torch.randn(c, generator=g) * 0.3).to(d)
    return mean, invstd, gamma, beta


def _bn_torch(x, mean, invstd, gamma, beta, relu=True):
    """bn.hip bn_apply, operation for operation, in torch fp32 (eager mode does not contract)."""
    t = (x.float() - mean) * invstd
    t = t * gamma
    t = t + beta
    if relu:
        t = torch.where(t > 0, t, torch.zeros_like(t))
    return t.bfloat16()


def _oracle_conv(x, w, tbl):
    xd, wd = x.double().cpu(), w.double().cpu()
    t = tbl.cpu().long()
    y = torch.zeros(t.shape[1], w.shape[2], dtype=torch.float64)
    for o in range(t.shape[0]):
        sel = t[o] >= 0
        y[sel] += xd[t[o][sel]] @ wd[o]
    return y


@pytest.mark.parametrize("m,kc,nc", [(40000, 16, 16), (40000, 32, 32), (9000, 32, 16), (9000, 16, 32), (777, 16, 16),
                                      (255, 32, 32), (5000, 16, 48)])
def test_prologue_kernel_is_bitwise_batchnorm_then_conv(native_lib, m, kc, nc):
    from doda_amd import ops
    d = dev()
    _, tbl = _scene_table(m, seed=31 + m)
    n = tbl.shape[1]
    torch.manual_seed(m + kc + nc)
    x = torch.randn(n, kc, device=d).bfloat16()
    w = (torch.randn(27, kc, nc, device=d) * 0.1).bfloat16().float()
    vec = _bn_vectors(kc, d, seed=m + kc)
    tb = ops.tilebook_build(tbl)
    assert ops.spconv_prologue_ok(kc, nc, 27, 2, False, n, n, True)
    z_ref = _bn_torch(x, *vec)
    y_ref = ops.spconv_gather(z_ref, w, tbl, n, 0, nc, tilebook=tb)
    z = torch.full_like(x, float("nan"))
    y = ops.spconv_gather(x, w, tbl, n, 0, nc, tilebook=tb, pre=(*vec, True, z))
    assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16))
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    # the definition: fp64 BatchNorm -> ReLU -> conv on the bf16-rounded normalised rows, one bf16 rounding of the output
    assert rel_err(y.float().cpu(), _oracle_conv(z_ref, w, tbl)) < 2.0 ** -7
    # without ReLU, without the side output
    z_lin = _bn_torch(x, *vec, relu=False)
    y_lin = ops.spconv_gather(x, w, tbl, n, 0, nc, tilebook=tb, pre=(*vec, False, None))
    assert torch.equal(y_lin, ops.spconv_gather(z_lin, w, tbl, n, 0, nc, tilebook=tb))
    # statistics epilogue + residual add ride along unchanged
    res = torch.randn(n, nc, device=d).bfloat16()
    ya, sa = ops.spconv_gather(x, w, tbl, n, 0, nc, tilebook=tb, residual=res, want_stats=True, pre=(*vec, True, z))
    yb, sb = ops.spconv_gather(z_ref, w, tbl, n, 0, nc, tilebook=tb, residual=res, want_stats=True)
    assert torch.equal(ya, yb) and torch.equal(sa, sb)


def test_prologue_on_tiles_without_a_list(native_lib):
    """Random neighbours: every tile exceeds the staged capacity and gathers through the dense table; absent
    neighbours must stay zero (NOT relu(beta - mean * ...)), the side output still covers every row."""
    from doda_amd import ops
    d = dev()
    n = 3000
    g = torch.Generator().manual_seed(5)
    tbl = torch.randint(0, n, (27, n), generator=g, dtype=torch.int32)
    tbl[torch.rand(27, n, generator=g) < 0.5] = -1
    tbl = tbl.to(d)
    tb = ops.tilebook_build(tbl)
    for kc in (16, 32):
        x = torch.randn(n, kc, device=d).bfloat16()
        w = (torch.randn(27, kc, 16, device=d) * 0.1).bfloat16().float()
        vec = _bn_vectors(kc, d, seed=kc)
        z_ref = _bn_torch(x, *vec)
        z = torch.full_like(x, float("nan"))
        y = ops.spconv_gather(x, w, tbl, n, 0, 16, tilebook=tb, pre=(*vec, True, z))
        assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16))
        assert torch.equal(y, ops.spconv_gather(z_ref, w, tbl, n, 0, 16, tilebook=tb))
        assert rel_err(y.float().cpu(), _oracle_conv(z_ref, w, tbl)) < 2.0 ** -7


def test_prologue_is_refused_where_it_is_not_built(native_lib):
    from doda_amd import ops
    from doda_amd._lib import DodaNativeError
    d = dev()
    _, tbl = _scene_table(9000, seed=2)
    n = tbl.shape[1]
    tb = ops.tilebook_build(tbl)
    assert not ops.spconv_prologue_ok(48, 48, 27, 2, False, n, n, True)      # no tile kernel for 48 channels
    assert not ops.spconv_prologue_ok(16, 16, 27, 2, False, n, n, False)     # no tilebook
    assert not ops.spconv_prologue_ok(16, 16, 27, 4, False, n, n, True)      # fp32 features
    assert not ops.spconv_prologue_ok(16, 16, 27, 2, True, n, n, True)       # fp32 output
    assert not ops.spconv_prologue_ok(16, 16, 8, 2, False, n, n // 2, True)  # not SubM
    for kc, kw in ((48, dict(tilebook=tb)), (16, dict())):
        x = torch.randn(n, kc, device=d).bfloat16()
        w = torch.randn(27, kc, 16, device=d) * 0.1
        with pytest.raises(DodaNativeError):
            ops.spconv_gather(x, w, tbl, n, 0, 16, pre=(*_bn_vectors(kc, d, 1), True, None), **kw)


@pytest.mark.parametrize("c,m", [(16, 50000), (32, 9001)])
def test_bn_fwd_final_matches_the_two_pass_batchnorm(native_lib, c, m):
    """doda_bn_fwd_final over the statistics rows of a conv epilogue: save_mean / save_invstd / running statistics as
    doda_bn_relu_fwd_stats computes them (same kernel), and close to torch's BatchNorm in fp64."""
    from doda_amd import ops
    d = dev()
    _, tbl = _scene_table(m, seed=c)
    n = tbl.shape[1]
    torch.manual_seed(c)
    x = torch.randn(n, c, device=d).bfloat16()
    w = (torch.randn(27, c, c, device=d) * 0.1)
    tb = ops.tilebook_build(tbl)
    y, stats = ops.spconv_gather(x, w, tbl, n, 0, c, tilebook=tb, want_stats=True)
    rm, rv = torch.zeros(c, device=d), torch.ones(c, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    mean, invstd = ops.bn_fwd_final(stats, n, 1e-4, 0.1, rm, rv, nbt)
    yd = y.double()
    mu, var = yd.mean(0), yd.var(0, unbiased=False)
    assert rel_err(mean.cpu(), mu.cpu()) < 1e-5
    assert rel_err(invstd.cpu(), (1.0 / torch.sqrt(var + 1e-4)).cpu()) < 1e-5
    assert rel_err(rm.cpu(), (0.1 * mu).cpu()) < 1e-5
    assert rel_err(rv.cpu(), (0.9 + 0.1 * yd.var(0, unbiased=True)).cpu()) < 1e-5
    assert int(nbt.item()) == 1


@pytest.mark.parametrize("skip", ["identity", "conv1x1"])
def test_residual_block_with_prologue_is_bitwise_the_two_launch_block(native_lib, skip):
    """model.ResidualBlock through ext.residual_block with the BatchNorm apply in the conv prologue against the same call
    with separate apply launches: output, statistics, every gradient and running statistic BIT-equal; and the block
    against torch's fp64 BatchNorm -> ReLU -> dense-definition conv."""
    ext = _ext_or_skip()
    from doda_amd import model as M
    from doda_amd import spconv
    from doda_amd.spconv import functional as Fsp
    d = dev()
    shape = [80, 70, 60]
    idx, _ = _scene_table(40000, seed=13)
    cin = 16 if skip == "identity" else 32
    torch.manual_seed(3)
    norm = lambda c: torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1)
    pre_conv = spconv.SubMConv3d(cin, cin, 3, padding=1, bias=False, indice_key="k").to(d)   # produces epilogue statistics
    blk = M.ResidualBlock(cin, 16, norm, indice_key="k").to(d)
    x0 = torch.randn(idx.shape[0], cin, device=d).bfloat16()
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(on):
        Fsp.set_bn_prologue(on)
        blk.load_state_dict(state)
        blk.train(); pre_conv.train()
        blk.zero_grad(set_to_none=True); pre_conv.zero_grad(set_to_none=True)
        xin = x0.clone().requires_grad_(True)
        inp = spconv.SparseConvTensor(xin, idx, shape, 2)
        data = spconv.ops.build_subm(idx, 2, shape, 3)
        data.tbl = ext.with_tilebook(data.tbl)
        inp.indice_dict["k"] = data
        out = blk(pre_conv(inp))
        out.features.float().square().mean().backward()
        torch.cuda.synchronize()
        return (out.features.detach().clone(), xin.grad.clone(), [p.grad.clone() for p in blk.parameters()],
                [b.clone() for b in blk.buffers()], pre_conv.weight.grad.clone())
    try:
        a = run(True)
        b = run(False)
    finally:
        Fsp.set_bn_prologue(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
    assert all(torch.equal(p, q) for p, q in zip(a[2], b[2]))
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))
    assert torch.isfinite(a[0].float()).all() and a[0].float().abs().max() > 0


def test_unet_step_with_and_without_prologue_is_bitwise(native_lib):
    """The bench's training step (two passes: weights repacked, running statistics moved) with the BatchNorm prologue on
    and off: loss, every gradient, every buffer BIT-equal; and the prologue really ran (fewer BatchNorm apply launches
    is checked by the profile; here: the switch is on and the tile levels carry tilebooks)."""
    _ext_or_skip()
    from doda_amd import model as M
    from doda_amd.scene import make_batch
    from doda_amd.spconv import functional as Fsp
    d = dev()
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(3, 60000, 17).items()}
    cfg = M.default_cfg()

    def run(on):
        assert Fsp.set_bn_prologue(on) == on
        torch.manual_seed(0)
        net = M.SparseConvNet(cfg).to(d).train()
        outs = []
        for _ in range(2):
            net.zero_grad(set_to_none=True)
            loss = M.cross_entropy(M.voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16), bd["labels"])
            loss.backward()
            outs.append(loss.detach().clone())
        torch.cuda.synchronize()
        return outs, [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]
    try:
        a = run(True)
        b = run(False)
    finally:
        Fsp.set_bn_prologue(False)
    assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
