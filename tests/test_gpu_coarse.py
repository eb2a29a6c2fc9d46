"""Coarse-level executor (csrc/coarse.hip, ABI 8): every op kind of doda_coarse_run against the oracle and against the
per-layer kernels it replaces, at the row / channel counts of the U-Net's levels 4-7 (reference model/unet_block.py:55-100;
SURVEY App. B: 16 i channels at level i).

The oracle restates spconv's indice_conv / indice_conv_backward (reference call sites model/unet_block.py:26,29,48,70,78)
in fp32 on the CPU; the executor computes in bf16 storage / fp32 accumulate like doda_spconv_gather_ex, so the comparisons
are made on bf16-rounded inputs with a bf16 output tolerance (2^-8 relative to the tensor's scale) and, against the
per-layer HIP kernel of the same arithmetic, to one bf16 unit in the last place of the tensor's scale.
"""
import numpy as np
import pytest
import torch

from tests.util import surface_voxels

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def _bf(t):
    return t.to(torch.bfloat16)


def _pack(w, K, kc, nc, layout, d):
    from doda_amd import ops
    plan = ops.PackPlan([(w, K, kc, nc, layout, 2)], d)
    plan.run()
    return plan.outputs[0]


def _scale_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


LEVELS = [(8400, 64), (1900, 80), (420, 96), (83, 112)]


def _level(seed, n, batch=4):
    side = max(16, int(round((n / batch / 0.08) ** (1 / 3))))   # ~8 % occupancy: 10-14 neighbours per voxel
    shape = [side, side, side]
    idx = surface_voxels(seed, n, batch, shape)
    return np.ascontiguousarray(idx[:n]), shape, batch


@pytest.mark.parametrize("n,c", LEVELS)
def test_gemm_subm_forward_vs_oracle_and_layer_kernel(native_lib, oracle, n, c):
    from doda_amd import ops
    d = dev()
    idx, shape, batch = _level(n, n)
    n = idx.shape[0]
    pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    g = torch.Generator().manual_seed(n)
    for cin, cout in ((c, c), (2 * c, c)):
        x = _bf(torch.randn(n, cin, generator=g)).to(d)
        res = _bf(torch.randn(n, cout, generator=g)).to(d)
        w = (torch.randn(27, cin, cout, generator=g) * (1.0 / (cin * 9)) ** 0.5).to(d)
        wp = _pack(w, 27, cin, cout, 0, d)
        G = ops.coarse_workgroups()
        y = torch.full((n, cout), float("nan"), dtype=torch.bfloat16, device=d)
        stats = torch.full((G, 2, cout), float("nan"), dtype=torch.float32, device=d)
        ops.coarse_run([dict(kind=ops.CX_GEMM, flags=0, rows=n, rows_in=n, c_in=cin, c_out=cout, K=27, tbl_ld=n, x_ld=cin, y_ld=cout,
                             res_ld=cout, x=x, w=wp, tbl=tbl, y=y, res=res, stats=stats)], d)
        torch.cuda.synchronize()
        assert not ops.coarse_error(d)
        # oracle on the bf16-rounded operands (weights are rounded to bf16 by the pre-pack)
        ref = oracle.indice_conv(x.float().cpu().numpy(), w.to(torch.bfloat16).float().cpu().numpy().reshape(3, 3, 3, cin, cout),
                                 pairs, pn, n, subm=True)
        ref = torch.as_tensor(ref) + res.float().cpu()
        assert _scale_err(y.float().cpu(), ref) < 2.0 ** -7, (cin, cout)
        lay = ops.spconv_gather(x, w.view(27, cin, cout), tbl, n, 0, cout, packed=wp, residual=res)
        assert _scale_err(y.float(), lay.float()) < 2.0 ** -7
        # statistics partials: column sums of the STORED tensor and of its squares
        yf = y.double()
        assert torch.allclose(stats[:, 0].double().sum(0), yf.sum(0), rtol=1e-4, atol=1e-3 * float(yf.abs().max()))
        assert torch.allclose(stats[:, 1].double().sum(0), (yf * yf).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n,c", LEVELS[1:])
def test_gemm_down_up_and_1x1_vs_layer_kernel(native_lib, oracle, n, c):
    """k2 s2 convolution (K = 8 tables, both directions) and the 1x1 skip convolution (identity table), strided operands."""
    from doda_amd import ops
    d = dev()
    idx, shape, batch = _level(7 * n, n)
    n = idx.shape[0]
    outids, child, par_off = ops.rulebook_down2(torch.from_numpy(idx).to(d), shape, batch)[:3]
    m = outids.shape[0]
    g = torch.Generator().manual_seed(n + 1)
    c2 = c + 16
    G = ops.coarse_workgroups()
    # strided conv fine -> coarse, reading x as the left half of a [n, 2c] matrix
    cat = _bf(torch.randn(n, 2 * c, generator=g)).to(d)
    w = (torch.randn(8, c, c2, generator=g) * (1.0 / (c * 4)) ** 0.5).to(d)
    wp = _pack(w, 8, c, c2, 0, d)
    y = torch.zeros((m, c2), dtype=torch.bfloat16, device=d)
    st = torch.zeros((G, 2, c2), dtype=torch.float32, device=d)
    ops.coarse_run([dict(kind=ops.CX_GEMM, flags=0, rows=m, rows_in=n, c_in=c, c_out=c2, K=8, tbl_ld=child.shape[1], x_ld=2 * c, y_ld=c2,
                         x=cat, w=wp, tbl=child, y=y, stats=st)], d)
    lay = ops.spconv_gather(cat[:, :c].contiguous(), w, child, m, 0, c2, packed=wp)
    assert _scale_err(y.float(), lay.float()) < 2.0 ** -7
    # inverse conv coarse -> fine, writing the right half of a [n, 2c] matrix
    z = _bf(torch.randn(m, c2, generator=g)).to(d)
    wi = (torch.randn(8, c2, c, generator=g) * (1.0 / c2) ** 0.5).to(d)
    wpi = _pack(wi, 8, c2, c, 0, d)
    out = torch.zeros((n, 2 * c), dtype=torch.bfloat16, device=d)
    ops.coarse_run([dict(kind=ops.CX_GEMM, flags=0, rows=n, rows_in=m, c_in=c2, c_out=c, K=8, tbl_ld=par_off.shape[1], x_ld=c2, y_ld=2 * c,
                         x=z, w=wpi, tbl=par_off, y=out[:, c:], stats=None)], d)
    lay = ops.spconv_gather(z, wi, par_off, n, 0, c, packed=wpi)
    assert _scale_err(out[:, c:].float(), lay.float()) < 2.0 ** -7
    assert float(out[:, :c].abs().max()) == 0.0
    # 1x1 convolution over the concatenation
    w1 = (torch.randn(1, 2 * c, c, generator=g) * (1.0 / (2 * c)) ** 0.5).to(d)
    wp1 = _pack(w1, 1, 2 * c, c, 0, d)
    s = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
    ops.coarse_run([dict(kind=ops.CX_GEMM, flags=ops.CX_F_IDENTITY, rows=n, rows_in=n, c_in=2 * c, c_out=c, K=1, tbl_ld=n, x_ld=2 * c, y_ld=c,
                         x=cat, w=wp1, tbl=None, y=s, stats=None)], d)
    ref = (cat.float() @ w1[0].to(torch.bfloat16).float())
    assert _scale_err(s.float(), ref) < 2.0 ** -7
    torch.cuda.synchronize()
    assert not ops.coarse_error(d)


@pytest.mark.parametrize("n,c", LEVELS[1:3])
def test_gemm_backward_epilogue_vs_definition(native_lib, oracle, n, c):
    """Data-gradient call: dz = (dy gathered through W[26-o]^T) * ReLU mask of the BatchNorm in front of the conv, statistics
    (sum dz, sum dz * xhat) of the stored values; also the 2c-channel output (two passes of channel blocks)."""
    from doda_amd import ops
    d = dev()
    idx, shape, batch = _level(3 * n, n)
    n = idx.shape[0]
    pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    g = torch.Generator().manual_seed(n + 2)
    G = ops.coarse_workgroups()
    for cin, cout in ((c, c), (2 * c, c)):
        x_bn = _bf(torch.randn(n, cin, generator=g)).to(d)            # input of the BatchNorm in front of the conv
        mean = (0.1 * torch.randn(cin, generator=g)).to(d)
        invstd = (1.0 + 0.2 * torch.rand(cin, generator=g)).to(d)
        gamma = (1.0 + 0.1 * torch.randn(cin, generator=g)).to(d)
        beta = (0.1 * torch.randn(cin, generator=g)).to(d)
        dy = _bf(torch.randn(n, cout, generator=g)).to(d)
        w = (torch.randn(27, cin, cout, generator=g) * (1.0 / (cin * 9)) ** 0.5).to(d)
        wpb = _pack(w, 27, cout, cin, 2, d)                           # data-grad layout of a SubM conv
        dz = torch.zeros((n, cin), dtype=torch.bfloat16, device=d)
        st = torch.zeros((G, 2, cin), dtype=torch.float32, device=d)
        ops.coarse_run([dict(kind=ops.CX_GEMM, flags=ops.CX_F_RELU, rows=n, rows_in=n, c_in=cout, c_out=cin, K=27, tbl_ld=n, x_ld=cout, y_ld=cin,
                             aux_ld=cin, x=dy, w=wpb, tbl=tbl, y=dz, aux=x_bn, stats=st, mean=mean, invstd=invstd, gamma=gamma, beta=beta)], d)
        torch.cuda.synchronize()
        assert not ops.coarse_error(d)
        xn = torch.relu((x_bn.float() - mean) * invstd * gamma + beta)
        din, _ = oracle.indice_conv_backward(xn.cpu().numpy(), w.to(torch.bfloat16).float().cpu().numpy().reshape(3, 3, 3, cin, cout),
                                             dy.float().cpu().numpy(), pairs, pn, subm=True)
        xh = (x_bn.float() - mean) * invstd
        mask = ((xh * gamma + beta) > 0).float()
        ref = torch.as_tensor(din).to(d) * mask
        assert _scale_err(dz.float(), ref) < 2.0 ** -7, (cin, cout)
        zf = dz.double()
        assert torch.allclose(st[:, 0].double().sum(0), zf.sum(0), rtol=1e-4, atol=1e-3 * float(zf.abs().max()))
        assert torch.allclose(st[:, 1].double().sum(0), (zf * xh.double()).sum(0), rtol=1e-4, atol=1e-3 * float(zf.abs().max()))


@pytest.mark.parametrize("n,c", LEVELS[1:])
def test_batchnorm_ops_vs_torch(native_lib, n, c):
    """STATS -> BNFWD (training, incl. the two-segment form of a concatenation, running statistics) and BNBWD (with the
    added skip gradient and the split output) against torch.nn.functional.batch_norm + autograd in fp32."""
    from doda_amd import ops
    import torch.nn.functional as F
    d = dev()
    g = torch.Generator().manual_seed(n + 3)
    G = ops.coarse_workgroups()
    C2 = 2 * c
    x = _bf(torch.randn(n, C2, generator=g) * 1.5 + 0.3).to(d)
    gamma = (1.0 + 0.1 * torch.randn(C2, generator=g)).to(d)
    beta = (0.1 * torch.randn(C2, generator=g)).to(d)
    rm, rv = torch.zeros(C2, device=d), torch.ones(C2, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    sa = torch.zeros((G, 2, c), dtype=torch.float32, device=d)
    sb = torch.zeros((G, 2, c), dtype=torch.float32, device=d)
    mean, invstd = torch.zeros(C2, device=d), torch.zeros(C2, device=d)
    y = torch.zeros((n, C2), dtype=torch.bfloat16, device=d)
    B = ops.CX_F_BARRIER
    ops.coarse_run([
        dict(kind=ops.CX_STATS, flags=0, rows=n, c_in=c, x_ld=C2, x=x, stats=sa),
        dict(kind=ops.CX_STATS, flags=0, rows=n, c_in=c, x_ld=C2, x=x[:, c:], stats=sb),
        dict(kind=ops.CX_BNFWD, flags=B | ops.CX_F_RELU | ops.CX_F_TRAINING, rows=n, c_in=C2, x_ld=C2, y_ld=C2, c_split=c, eps=1e-4, momentum=0.1,
             x=x, y=y, stats=sa, stats_b=sb, gamma=gamma, beta=beta, mean=mean, invstd=invstd, running_mean=rm, running_var=rv, nbt=nbt),
    ], d)
    xf = x.float().requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(C2, device=d), torch.ones(C2, device=d)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = torch.relu(F.batch_norm(xf, rm_ref, rv_ref, gr, br, True, 0.1, 1e-4))
    torch.cuda.synchronize()
    assert not ops.coarse_error(d)
    assert _scale_err(y.float(), y_ref.detach()) < 2.0 ** -7
    assert torch.allclose(rm, rm_ref, rtol=1e-4, atol=1e-5) and torch.allclose(rv, rv_ref, rtol=1e-4, atol=1e-5)
    assert int(nbt) == 1
    assert torch.allclose(mean, xf.detach().mean(0), rtol=1e-4, atol=1e-5)
    # backward: the masked gradient dz and its statistics as the GEMM epilogue delivers them
    dy = _bf(torch.randn(n, C2, generator=g)).to(d)
    add = _bf(torch.randn(n, C2, generator=g)).to(d)
    (y_ref * dy.float()).sum().backward()
    xh = (x.float() - mean) * invstd
    dz = _bf(dy.float() * ((xh * gamma + beta) > 0).float())
    st = torch.zeros((G, 2, C2), dtype=torch.float32, device=d)
    st[0, 0] = dz.float().sum(0)
    st[0, 1] = (dz.float() * xh).sum(0)
    dxa = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
    dxb = torch.zeros((n, c), dtype=torch.bfloat16, device=d)
    dg, db = torch.zeros(C2, device=d), torch.zeros(C2, device=d)
    ops.coarse_run([dict(kind=ops.CX_BNBWD, flags=0, rows=n, c_in=C2, x_ld=C2, aux_ld=C2, res_ld=C2, y_ld=c, y2_ld=c, c_split=c,
                         x=dz, aux=x, res=add, y=dxa, y2=dxb, stats=st, mean=mean, invstd=invstd, gamma=gamma, dgamma=dg, dbeta=db)], d)
    torch.cuda.synchronize()
    assert not ops.coarse_error(d)
    ref = xf.grad + add.float()
    got = torch.cat((dxa, dxb), 1).float()
    assert _scale_err(got, ref) < 2.0 ** -6
    assert torch.allclose(dg, gr.grad, rtol=2e-2, atol=2e-2 * float(gr.grad.abs().max()))
    assert torch.allclose(db, br.grad, rtol=2e-2, atol=2e-2 * float(br.grad.abs().max()))
    # evaluation mode: running statistics
    ye = torch.zeros((n, C2), dtype=torch.bfloat16, device=d)
    ops.coarse_run([dict(kind=ops.CX_BNFWD, flags=ops.CX_F_RELU, rows=n, c_in=C2, x_ld=C2, y_ld=C2, c_split=C2, eps=1e-4, momentum=0.1,
                         x=x, y=ye, gamma=gamma, beta=beta, running_mean=rm, running_var=rv)], d)
    ye_ref = torch.relu(F.batch_norm(x.float(), rm, rv, gamma, beta, False, 0.1, 1e-4))
    assert _scale_err(ye.float(), ye_ref) < 2.0 ** -7


def test_chain_with_barriers_is_repeatable(native_lib):
    """A residual block's forward as six dependent ops in ONE launch (BNFWD -> GEMM -> BNFWD -> GEMM + skip), twice: bit-equal
    results (fixed summation orders, no atomics) and equal to the same chain issued op by op (one launch each)."""
    from doda_amd import ops
    d = dev()
    n, c = 1900, 80
    idx, shape, batch = _level(99, n)
    n = idx.shape[0]
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    g = torch.Generator().manual_seed(7)
    G = ops.coarse_workgroups()
    x = _bf(torch.randn(n, c, generator=g)).to(d)
    w1 = (torch.randn(27, c, c, generator=g) * (1.0 / (c * 9)) ** 0.5).to(d)
    w2 = (torch.randn(27, c, c, generator=g) * (1.0 / (c * 9)) ** 0.5).to(d)
    p1, p2 = _pack(w1, 27, c, c, 0, d), _pack(w2, 27, c, c, 0, d)
    ga, be = torch.ones(c, device=d), torch.zeros(c, device=d)

    def run(fused):
        s0, s1, s2 = (torch.zeros((G, 2, c), dtype=torch.float32, device=d) for _ in range(3))
        a1, y1, a2, y2 = (torch.zeros((n, c), dtype=torch.bfloat16, device=d) for _ in range(4))
        m1, i1, m2, i2 = (torch.zeros(c, device=d) for _ in range(4))
        B = ops.CX_F_BARRIER
        T = ops.CX_F_RELU | ops.CX_F_TRAINING
        chain = [
            dict(kind=ops.CX_STATS, flags=0, rows=n, c_in=c, x_ld=c, x=x, stats=s0),
            dict(kind=ops.CX_BNFWD, flags=B | T, rows=n, c_in=c, x_ld=c, y_ld=c, c_split=c, eps=1e-4, momentum=0.1, x=x, y=a1, stats=s0,
                 gamma=ga, beta=be, mean=m1, invstd=i1),
            dict(kind=ops.CX_GEMM, flags=B, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, x=a1, w=p1, tbl=tbl, y=y1, stats=s1),
            dict(kind=ops.CX_BNFWD, flags=B | T, rows=n, c_in=c, x_ld=c, y_ld=c, c_split=c, eps=1e-4, momentum=0.1, x=y1, y=a2, stats=s1,
                 gamma=ga, beta=be, mean=m2, invstd=i2),
            dict(kind=ops.CX_GEMM, flags=B, rows=n, rows_in=n, c_in=c, c_out=c, K=27, tbl_ld=n, x_ld=c, y_ld=c, res_ld=c, x=a2, w=p2, tbl=tbl, y=y2,
                 res=x, stats=s2),
        ]
        if fused:
            ops.coarse_run(chain, d)
        else:
            for o in chain:
                ops.coarse_run([o], d)
        torch.cuda.synchronize()
        assert not ops.coarse_error(d)
        return y2.clone(), s2.clone()

    ya, sa = run(True)
    yb, sb = run(True)
    yc, sc = run(False)
    assert torch.equal(ya, yb) and torch.equal(sa, sb)
    assert torch.equal(ya, yc) and torch.equal(sa, sc)
    # against the per-layer kernels
    import torch.nn.functional as F
    a1 = _bf(torch.relu(F.batch_norm(x.float(), None, None, ga, be, True, 0.1, 1e-4)))
    y1 = ops.spconv_gather(a1, w1, tbl, n, 0, c, packed=p1)
    a2 = _bf(torch.relu(F.batch_norm(y1.float(), None, None, ga, be, True, 0.1, 1e-4)))
    y2 = ops.spconv_gather(a2, w2, tbl, n, 0, c, packed=p2, residual=x)
    assert _scale_err(ya.float(), y2.float()) < 2.0 ** -6
