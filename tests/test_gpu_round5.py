"""Round 5 / ABI 9: BatchNorm statistics of the conv epilogues as fp64 TOTALS (doda_conv_epilogue.stats_totals: hardware fp64
atomics into 8 slots x 2 sums x nc / 4 padded groups of doubles) and the one-launch BatchNorm over them (doda_bn_relu_fwd_totals / _bwd_totals), against the
per-workgroup rows + reduction launch they replace (reference: torch.nn.BatchNorm1d + ReLU applied by SparseSequential,
model/unet_block.py:23-30,46-49; spconv indice_conv, model/unet_block.py:26,29)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene(oracle):
    from tests.test_gpu_round4 import _big_scene, _hip_rulebook
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    return idx.shape[0], tbl, tb


@pytest.mark.parametrize("cin,cout,tiled", [(16, 16, True), (32, 32, True), (16, 16, False), (64, 32, False), (48, 48, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_epilogue_totals_equal_the_sum_of_the_epilogue_rows(native_lib, scene, cin, cout, tiled, dtype):
    """Same call, statistics once as rows and once as totals: y bit-equal, and the totals (summed over their eight slots)
    equal the fp64 sum of the rows to fp64 rounding — for the tile kernels (conv_tile16 / conv_tile with 1250 tiles, incl. tiles
    without a list), conv_fast and conv_wlds48; forward sums and, with the BatchNorm operands, the backward sums of a
    data-gradient call.  Also against the fp64 column sums of the stored output."""
    from doda_amd import ops
    if tiled and dtype != torch.bfloat16:
        pytest.skip("tile kernels: bf16 rows")
    n, tbl, tb = scene
    d = dev()
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(n, cin, generator=g).to(dtype).to(d)
    w = (torch.randn(27, cin, cout, generator=g) * 0.1).to(d)
    res = torch.randn(n, cout, generator=g).to(dtype).to(d)
    kw = dict(tilebook=tb if tiled else None, residual=res)
    y0, rows = ops.spconv_gather(x, w, tbl, n, 0, cout, want_stats=True, **kw)
    y1, tot = ops.spconv_gather(x, w, tbl, n, 0, cout, want_stats="totals", **kw)
    assert torch.equal(y0, y1) and tot.dtype == torch.float64 and tuple(tot.shape) == (8, 2, cout // 4, 16)
    assert float(tot[..., 4:].abs().max()) == 0.0            # (the padding of the 128-byte lines stays untouched)
    want = rows.double().sum(0)
    got = ops.totals_sums(tot)
    assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())
    col = y1.double()
    ref = torch.stack([col.sum(0), (col * col).sum(0)])
    assert float((got - ref).abs().max()) <= (2e-3 if dtype == torch.bfloat16 else 2e-5) * float(ref.abs().max())
    # accumulating into totals that already hold something: the sums add
    y2, tot2 = ops.spconv_gather(x, w, tbl, n, 0, cout, want_stats=tot.clone(), **kw)
    assert float((ops.totals_sums(tot2) - 2 * want).abs().max()) <= 1e-12 * float(want.abs().max())
    # backward sums (sum dz, sum dz * xhat) of a data-gradient call
    bx = torch.randn(n, cin, generator=g).to(dtype).to(d)
    mean = torch.randn(cin, generator=g).to(d) * 0.1
    invstd = (torch.rand(cin, generator=g) + 0.5).to(d)
    gamma = (torch.rand(cin, generator=g) + 0.5).to(d)
    beta = (torch.randn(cin, generator=g) * 0.1).to(d)
    dy = torch.randn(n, cout, generator=g).to(dtype).to(d)
    wb = (torch.randn(27, cout, cin, generator=g) * 0.1).to(d)
    kwb = dict(tilebook=tb if tiled else None, bn=(bx, mean, invstd, gamma, beta, True))
    d0, rows_b = ops.spconv_gather(dy, wb, tbl, n, 0, cin, want_stats=True, **kwb)
    d1, tot_b = ops.spconv_gather(dy, wb, tbl, n, 0, cin, want_stats="totals", **kwb)
    assert torch.equal(d0, d1)
    want_b = rows_b.double().sum(0)
    assert float((ops.totals_sums(tot_b) - want_b).abs().max()) <= 1e-12 * float(want_b.abs().max()) + 1e-300


@pytest.mark.parametrize("c", [16, 32, 64, 96])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batchnorm_over_totals_equals_batchnorm_over_rows(native_lib, c, dtype):
    """doda_bn_relu_fwd_totals / _bwd_totals against doda_bn_relu_fwd_stats / _bwd_stats on statistics of the same tensor: the
    totals are the rows' fp64 sums spread over eight slots, so mean / invstd / running statistics / dgamma / dbeta and the swept
    tensors must be bit-equal; a channel concatenation (two producers) against the two-reductions + apply sequence; the strided
    second-gradient operand."""
    from doda_amd import ops
    d = dev()
    m = 70001
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(m, c, generator=g) * 2 + 0.3).to(dtype).to(d)
    gamma = (torch.rand(c, generator=g) + 0.5).to(d)
    beta = (torch.randn(c, generator=g) * 0.1).to(d)
    xs = x.float()
    parts = 37
    edges = torch.linspace(0, m, parts + 1).long()
    rows = torch.stack([torch.stack([xs[a:b].sum(0), (xs[a:b] * xs[a:b]).sum(0)]) for a, b in zip(edges[:-1], edges[1:])])   # [37, 2, c] fp32
    tot = ops.totals_from_rows(rows)

    def buffers():
        return torch.zeros(c, device=d), torch.ones(c, device=d), torch.zeros((), dtype=torch.int64, device=d)
    rm0, rv0, nb0 = buffers()
    rm1, rv1, nb1 = buffers()
    mu0, is0 = ops.bn_fwd_final(rows.contiguous(), m, 1e-4, 0.1, rm0, rv0, nb0)     # (the reduction launch of the rows form)
    y1, mu1, is1 = ops.bn_relu_fwd_totals(x, tot, gamma, beta, rm1, rv1, 0.1, 1e-4, True, nb1)
    assert torch.equal(mu0, mu1) and torch.equal(is0, is1) and torch.equal(rm0, rm1) and torch.equal(rv0, rv1) and int(nb1) == 1
    ref = torch.relu((xs - mu1) * is1 * gamma + beta)
    assert float((y1.float() - ref).abs().max()) <= (2e-2 if dtype == torch.bfloat16 else 1e-5) * float(ref.abs().max())
    # a concatenation [a | b]: two producers' totals
    ca = c // 2 if (c // 2) % 4 == 0 else 16
    if 0 < ca < c:
        ta = tot[:, :, :ca // 4].contiguous()
        tb_ = tot[:, :, ca // 4:].contiguous()
        rm2, rv2, nb2 = buffers()
        y2, mu2, is2 = ops.bn_relu_fwd_totals(x, ta, gamma, beta, rm2, rv2, 0.1, 1e-4, True, nb2, totals_b=tb_)
        assert torch.equal(mu2, mu1) and torch.equal(is2, is1) and torch.equal(y2, y1) and torch.equal(rm2, rm1)
    # backward
    dy = torch.randn(m, c, generator=g).to(dtype).to(d)
    xh = (xs - mu1) * is1
    dz = torch.where(xh * gamma + beta > 0, dy.float(), torch.zeros((), device=d))
    rows_b = torch.stack([torch.stack([dz[a:b].sum(0), (dz[a:b] * xh[a:b]).sum(0)]) for a, b in zip(edges[:-1], edges[1:])])
    tot_b = ops.totals_from_rows(rows_b)
    wide = torch.randn(m, 2 * c, generator=g).to(dtype).to(d)
    for add in (None, torch.randn(m, c, generator=g).to(dtype).to(d), wide[:, c:]):
        dx0, dg0, db0 = ops.bn_relu_bwd_stats(x, dy, rows_b.contiguous(), mu1, is1, gamma, beta, True, add=add)
        dx1, dg1, db1 = ops.bn_relu_bwd_totals(x, dy, tot_b, mu1, is1, gamma, beta, True, add=add)
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(dx0, dx1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unet_step_with_totals_equals_the_step_with_rows(native_lib, dtype):
    """The U-Net training step with the statistics as totals (default) against rows + reduction launches: the statistics differ
    at most by fp64 rounding of their totals, so loss and every parameter gradient agree to fp32 accumulation noise; and the step
    launches fewer kernels."""
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch
    from doda_amd.spconv import functional as Fsp
    from tests.util import deterministic_init
    ext = Fsp._ext
    if ext is None or not hasattr(ext, "set_stats_totals"):
        pytest.skip("extension without the totals switch")
    d = dev()
    cfg = default_cfg()
    b = make_batch(2, 60000, 31)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    got = {}
    was = ext.get_stats_totals()
    from doda_amd import model as M
    old_mode = (M.COARSE_MODE, M.COARSE_EXEC_LEVEL)
    M.set_coarse_mode("off")       # (the op list of the default path keeps its statistics as totals whatever the switch says)
    assert Fsp.set_deferred_wgrad(True)
    try:
        for on in (True, False):
            ext.set_stats_totals(on)
            net = deterministic_init(SparseConvNet(cfg), seed=6).to(d).train()
            loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=dtype), bd["labels"], ignore_index=255)
            loss.backward()
            torch.cuda.synchronize()
            got[on] = (float(loss), {k: p.grad.detach().float().clone() for k, p in net.named_parameters()},
                       {k: v.detach().clone() for k, v in net.named_buffers()})
    finally:
        ext.set_stats_totals(was)
        Fsp.set_deferred_wgrad(False)
        M.set_coarse_mode(*old_mode)
    (l1, g1, b1), (l0, g0, b0) = got[True], got[False]
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert abs(l0 - l1) <= 1e-5 * abs(l0) + (1e-3 if dtype == torch.bfloat16 else 0.0)
    for k, a in g0.items():
        assert float((a - g1[k]).norm()) <= tol * float(a.norm()) + 1e-7, k
    for k, a in b0.items():
        assert float((a.double() - b1[k].double()).abs().max()) <= 1e-5 * float(a.double().abs().max()) + 1e-6, k


def test_tile_kernels_on_a_two_million_voxel_batch(native_lib):
    """BASELINE config 5 at batch 4 in the loader's Z-order numbering: 2.0 M voxels, 7770 tiles (15 per conv_tile16 workgroup; the
    unit tests above stop at 1250), a level-1 rulebook over 8.5e7 cells (direct-address grid since the limit is 2^28; the hash
    builder must give the same table), tiles whose rows span more than the tilebook builder's bitmap covers.  No CPU oracle at this size:
    conv_tile16 against the plain conv_tile bit for bit (forward and data gradient in the step's form, statistics as totals),
    both against the dense-table kernel to bf16 output rounding, the tile weight gradient against the gather-table kernel, and
    everything twice (a second run must reproduce the first exactly)."""
    from doda_amd import ops, spconv
    from doda_amd._lib import lib
    from doda_amd.collate import reorder_voxels
    from doda_amd.scene import make_batch
    d = dev()
    b = reorder_voxels(make_batch(4, 500000, 1000, 100), "morton")
    idx = b["voxel_locs"].int().to(d)
    shape = [int(s) for s in b["spatial_shape"]]
    n = idx.shape[0]
    assert n > 1900000 and 4 * shape[0] * shape[1] * shape[2] > (1 << 26)
    tbl = ops.rulebook_subm(idx, shape, 4, 3)
    import ctypes as C
    from doda_amd._lib import check
    ws = torch.empty(lib().doda_rulebook_workspace_bytes(n), dtype=torch.uint8, device=d)        # minimum workspace: the hash builder
    tbl_h = torch.empty_like(tbl)
    check(lib().doda_rulebook_subm(idx.data_ptr(), n, (C.c_int32 * 3)(*shape), 4, 3, tbl_h.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream().cuda_stream), "doda_rulebook_subm")
    assert torch.equal(tbl, tbl_h)
    del tbl_h, ws
    tb = ops.tilebook_build(tbl)
    n_over = tb[-8:].view(torch.int32).cpu().tolist()
    assert n_over[1] == 0                                   # (the renumbering's point: every tile keeps its list)
    # the builder gives the same bytes every time — 131 of this table's tiles take its hash + bitonic-sort form, whose barrier
    # elision (round 3) corrupted a list in ~1 of 400 builds (the rare wrong steps of 1 cm training runs)
    torch.cuda.synchronize()
    tb_ref = tb.clone()
    differ = sum(int(not torch.equal(ops.tilebook_build(tbl), tb_ref)) for _ in range(2500))
    assert differ == 0, "%d of 2500 tilebook builds differ from the first" % differ
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, 16, generator=g).bfloat16().to(d)
    dy = torch.randn(n, 16, generator=g).bfloat16().to(d)
    res = torch.randn(n, 16, generator=g).bfloat16().to(d)
    w = (torch.randn(27, 16, 16, generator=g) * 0.1).to(d)
    runs = []
    for rep in range(2):
        out = {}
        for on in (1, 0):
            assert lib().doda_set_option(4, on) == 0        # DODA_OPT_TILE_PIPELINE: conv_tile16 / plain conv_tile
            try:
                out[on] = [ops.spconv_gather(inp, w, tbl, n, layout, 16, tilebook=tb, residual=res, want_stats="totals")
                           for inp, layout in ((x, 0), (dy, 2))]
            finally:
                lib().doda_set_option(4, 1)
        for (y1, t1), (y0, t0) in zip(out[1], out[0]):
            assert torch.equal(y1, y0)
            s1, s0 = ops.totals_sums(t1), ops.totals_sums(t0)
            assert float((s1 - s0).abs().max()) <= 1e-6 * float(s0.abs().max())      # (fp32 partials of 512 against 768 workgroups)
        dense = ops.spconv_gather(x, w, tbl, n, 0, 16, residual=res)
        assert (dense != out[1][0][0]).float().mean().item() < 0.02 and rel_err_t(out[1][0][0], dense) < 2.0 ** -6
        dw_t, = ops.spconv_wgrad_multi([(x, dy, tbl, n, None, None, tb)])
        dw_g, = ops.spconv_wgrad_multi([(x, dy, tbl, n)])
        assert float((dw_t - dw_g).norm()) <= 2e-3 * float(dw_g.norm())
        runs.append((out[1][0][0], out[1][1][0], ops.totals_sums(out[1][0][1]), dw_t))
    for a, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a, b_)


def rel_err_t(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("ca,cb,c_out", [(3, 3, 16), (3, 3, 8), (3, 0, 4), (6, 0, 6), (1, 2, 3)])
def test_input_rows_in_one_launch_equal_pool_cast_pad(native_lib, oracle, mode, ca, cb, c_out):
    """doda_voxelize_fp_rows (cat + voxel pooling + cast + channel padding in ONE launch, reference model/unet.py:89-94): the
    fp32 form bit for bit the oracle's voxelize_fp of the concatenated features (restated from voxelize.cu:10-31) with zero
    channels behind it; the bf16 form that result rounded to nearest even (torch's cast)."""
    from doda_amd import ops
    from tests.test_gpu_parity import _points
    coords = _points(22, 30000, 3, 24)
    _, _, om = oracle.voxelize_idx(coords, mode)
    rng = np.random.default_rng(ca * 10 + cb)
    fa = rng.standard_normal((coords.shape[0], ca)).astype(np.float32) * 3
    fb = rng.standard_normal((coords.shape[0], cb)).astype(np.float32) if cb else None
    ref = oracle.voxelize_fp(np.ascontiguousarray(np.concatenate([fa, fb], 1)) if cb else fa, om, average=(mode == 4))
    want = np.zeros((om.shape[0], c_out), np.float32)
    want[:, :ca + cb] = ref
    d = dev()
    rules = torch.from_numpy(om).to(d)
    ta, tb = torch.from_numpy(fa).to(d), (torch.from_numpy(fb).to(d) if cb else None)
    got = ops.voxelize_fp_rows(ta, tb, rules, mode, c_out, torch.float32)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    got16 = ops.voxelize_fp_rows(ta, tb, rules, mode, c_out, torch.bfloat16)
    assert torch.equal(got16.cpu().view(torch.int16), torch.from_numpy(want).to(torch.bfloat16).view(torch.int16))
    with pytest.raises(RuntimeError):
        ops.voxelize_fp_rows(ta, tb, rules, mode, ca + cb - 1, torch.float32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unet_step_with_the_one_launch_input_equals_the_four_launch_input(native_lib, dtype):
    """voxelize_and_run with the input layer's rows written by doda_voxelize_fp_rows against torch.cat + voxelization + cast +
    pad_channels: the same rows, so the same loss and the same gradients, bit for bit."""
    from doda_amd import model as M
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch
    from tests.util import deterministic_init
    d = dev()
    cfg = default_cfg()
    b = make_batch(2, 40000, 37)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    got = {}
    was = M.INPUT_ROWS
    try:
        for on in (True, False):
            M.INPUT_ROWS = on
            net = deterministic_init(SparseConvNet(cfg), seed=8).to(d).train()
            use_xyz = bool(cfg.MODEL.BACKBONE.use_xyz)
            c_in = 6 if use_xyz else 3
            rows = M._input_rows(net, bd["feats"], bd["locs_float"] if use_xyz else None, bd["v2p_map"], 4, dtype)
            assert (rows is not None) == on
            if on:
                assert rows.shape[1] == (16 if dtype == torch.bfloat16 else c_in + (-c_in) % 4) and rows._doda_padded_from == c_in
                assert M._input_rows(net, bd["feats"], bd["locs_float"] if not use_xyz else None, bd["v2p_map"], 4, dtype) is None
            loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=dtype), bd["labels"], ignore_index=255)
            loss.backward()
            torch.cuda.synchronize()
            got[on] = (float(loss), {k: p.grad.detach().float().clone() for k, p in net.named_parameters()})
    finally:
        M.INPUT_ROWS = was
    assert got[True][0] == got[False][0]
    for k, a in got[False][1].items():
        assert torch.equal(a, got[True][1][k]), k
