"""Round-2 GPU parity tests.

* packed conv weights follow EVERY kind of weight update (VERDICT r1 "stale packed weights"): three
  optimizer steps of the U-Net with SGD(fused=True), SGD(foreach=True) and raw `.data` edits must give
  the same step-3 loss and gradients as the path that packs inside every call, and as the CPU oracle
  U-Net stepped the same way;
* the pair-list weight gradient (doda_spconv_wgrad_pairs_bf16) against the oracle's
  indice_conv_backward on the SAME bf16-rounded operands (products are then exact in fp32, only the
  summation order differs: 1e-4 rel as north_star states for fp32 features);
* deferred weight gradients: a second backward pass accumulates, a weight used twice in one graph gets
  both contributions, an aborted backward leaves nothing behind."""
import os

import numpy as np
import pytest
import torch

from tests.util import deterministic_init, surface_voxels

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------ weights that change under the pack cache
def _three_steps_hip(style, prepack, batch):
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.spconv import conv as dconv
    d = dev()
    cfg = default_cfg()
    net = deterministic_init(SparseConvNet(cfg), seed=0).to(d).train()
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items()}
    dconv.set_prepack(prepack)
    try:
        return _three_steps(net, style, lambda: cross_entropy(
            voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.float32), bd["labels"]), d), dconv
    finally:
        dconv.set_prepack(True)


def _three_steps(net, style, loss_fn, d):
    lr = 0.2
    opt = None
    if style == "fused":
        opt = torch.optim.SGD(net.parameters(), lr=lr, fused=True)
    elif style == "foreach":
        opt = torch.optim.SGD(net.parameters(), lr=lr, foreach=True)
    losses, grads = [], None
    for step in range(3):
        net.zero_grad(set_to_none=True)
        loss = loss_fn()
        loss.backward()
        losses.append(float(loss))
        if step == 2:
            grads = {k: p.grad.detach().double().cpu().numpy() for k, p in net.named_parameters()}
        if opt is not None:
            opt.step()
        else:  # "data": raw edits that bump no version counter
            for p in net.parameters():
                p.data.add_(p.grad, alpha=-lr)
    return losses, grads


@pytest.mark.parametrize("style", ["fused", "foreach", "data"])
def test_packed_weights_follow_every_update_style(native_lib, style):
    from doda_amd.scene import make_batch
    from oracle.unet_cpu import OracleUNet, forward_backward
    batch = make_batch(2, 3000, 77)
    (l_pack, g_pack), dconv = _three_steps_hip(style, True, batch)
    (l_nopack, g_nopack), _ = _three_steps_hip(style, False, batch)
    for a, b in zip(l_pack, l_nopack):
        assert abs(a - b) <= 1e-5 * abs(b), (style, l_pack, l_nopack)
    for k in g_pack:
        assert rel_err(g_pack[k], g_nopack[k]) < 1e-4, (style, k)
    # the check has teeth: with the packed copies frozen at their step-0 values (the round-1 behaviour
    # under a fused optimizer) the step-3 loss is far outside that tolerance
    orig = dconv._repack_all
    done = {}

    def frozen(device, esz):
        if (device, esz) not in done:
            done[(device, esz)] = orig(device, esz)
        plan = done[(device, esz)]
        plan.gen = dconv._GEN[0]
        plan.versions = tuple(m().weight._version for m in plan.refs)
        return plan
    dconv._repack_all = frozen
    try:
        (l_stale, _), _ = _three_steps_hip(style, True, batch)
    finally:
        dconv._repack_all = orig
        dconv.invalidate_packed()
    assert abs(l_stale[0] - l_pack[0]) <= 1e-5 * abs(l_pack[0])
    assert abs(l_stale[2] - l_pack[2]) > 1e-2 * abs(l_pack[2]), (l_stale, l_pack)
    # CPU oracle U-Net (fp64) stepped the same way.  fp32 against fp64 drifts apart over optimizer steps:
    # the network's deep levels normalise ~40 rows per channel with 1/sqrt(var + 1e-4) up to 100, which
    # amplifies rounding (tools/graddiag.py: step-1 loss equal to 1e-7, gradients to 1e-2); the bounds
    # below are ~3x the measured drift and far below the stale-weight deviation asserted above.
    net = deterministic_init(OracleUNet(), seed=0).double().train()
    l_ref, g_ref = _three_steps(net, "foreach" if style == "fused" else style,
                                lambda: forward_backward_loss(net, batch), torch.device("cpu"))
    for a, b, tol in zip(l_pack, l_ref, (1e-5, 3e-3, 6e-2)):
        assert abs(a - b) <= tol * abs(b), (style, l_pack, l_ref)
    assert abs(l_stale[2] - l_ref[2]) > 2 * abs(l_pack[2] - l_ref[2]), (l_stale, l_pack, l_ref)


def forward_backward_loss(net, batch):
    """Loss of the oracle U-Net (the caller runs backward)."""
    from oracle import oracle as orc
    from oracle import spconv_cpu as sp
    vf = torch.from_numpy(orc.voxelize_fp(batch["feats"].numpy(), batch["v2p_map"].numpy(), True))
    vf = vf.to(next(net.parameters()).dtype)
    inp = sp.SparseConvTensor(vf, batch["voxel_locs"].int(), batch["spatial_shape"], batch["offsets"].numel() - 1)
    return torch.nn.functional.cross_entropy(net(inp, batch["p2v_map"]), batch["labels"], ignore_index=255)


def test_pack_runs_once_per_forward_and_after_optimizer_steps(native_lib):
    """The generation protocol: one re-pack per forward pass, none inside it."""
    from doda_amd import ops, spconv
    from doda_amd.spconv import conv as dconv
    d = dev()
    calls = []
    orig = ops.PackPlan.run
    ops.PackPlan.run = lambda self: (calls.append(1), orig(self))[1]
    try:
        torch.manual_seed(0)
        c1 = spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="k").to(d)
        c2 = spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="k").to(d)
        idx = torch.from_numpy(surface_voxels(1, 2000, 1, [30, 30, 30])).to(d)
        x = torch.randn(idx.shape[0], 16, device=d)
        opt = torch.optim.SGD(list(c1.parameters()) + list(c2.parameters()), lr=0.1, fused=True)
        dconv.invalidate_packed()
        n0 = len(calls)
        for _ in range(3):
            st = spconv.SparseConvTensor(x, idx, [30, 30, 30], 1)
            y = c2(c1(st)).features
            y.square().mean().backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        assert len(calls) - n0 == 3, len(calls) - n0
        # same input tensor again, weights untouched: no re-pack; after an in-place edit: re-pack
        n1 = len(calls)
        st = spconv.SparseConvTensor(x, idx, [30, 30, 30], 1)
        a = c1(st).features
        b = c1(st).features
        assert len(calls) - n1 == 1 and torch.equal(a, b)
        with torch.no_grad():
            c1.weight.mul_(2.0)
        c = c1(st).features
        assert len(calls) - n1 == 2
        assert torch.allclose(c, 2 * a, rtol=1e-5, atol=1e-6)
    finally:
        ops.PackPlan.run = orig


# ------------------------------------------------------------------ pair-list weight gradient
def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16)


def _subm_case(oracle, seed, n, shape, batch):
    idx = surface_voxels(seed, n, batch, shape)
    pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
    return idx, pairs, pn


@pytest.mark.parametrize("cin,cout,n", [(16, 16, 30000), (32, 16, 9000), (48, 48, 4000), (112, 112, 700),
                                        (16, 32, 5000), (64, 80, 1500)])
def test_wgrad_pairs_subm_vs_oracle(native_lib, oracle, cin, cout, n):
    from doda_amd import ops
    shape, batch = [40, 36, 30], 2
    idx, pairs, pn = _subm_case(oracle, cin + cout, n, shape, batch)
    m = idx.shape[0]
    rng = np.random.default_rng(cin * 7 + cout)
    xb = _bf16_round(rng.standard_normal((m, cin)).astype(np.float32))
    gb = _bf16_round(rng.standard_normal((m, cout)).astype(np.float32))
    w64 = torch.zeros(3, 3, 3, cin, cout, dtype=torch.float64)
    _, ref_dw = oracle.indice_conv_backward(xb.double(), w64, gb.double(), pairs, pn, False, True)
    d = dev()
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    pr, num, seg = ops.rulebook_pairs(tbl, m, flip=True, pad=False, with_seg=True)
    assert np.array_equal(num.cpu().numpy(), pn)
    # the segment prefix: pairs of list o whose input row lies below tile t
    tile = 256
    in_rows = pairs[0]
    want = np.stack([[int((in_rows[o, :pn[o]] < t * tile).sum()) for t in range(seg.shape[1])] for o in range(27)])
    assert np.array_equal(seg.cpu().numpy(), want)
    dw = ops.spconv_wgrad_pairs(xb.to(d), gb.to(d), pr[0], pr[1], num, seg)
    assert rel_err(dw.cpu().reshape(ref_dw.shape), ref_dw) < RTOL
    # the gather-table kernel computes the same thing
    dw_tbl = ops.spconv_wgrad(xb.to(d), gb.to(d), tbl, m)
    assert rel_err(dw.cpu(), dw_tbl.cpu()) < RTOL
    # accumulate: dw += result
    base = torch.randn_like(dw)
    acc = ops.spconv_wgrad_pairs(xb.to(d), gb.to(d), pr[0], pr[1], num, seg, accumulate_into=base.clone())
    assert rel_err((acc - base).cpu(), dw.cpu()) < 1e-5
    # bitwise repeatable
    assert torch.equal(dw, ops.spconv_wgrad_pairs(xb.to(d), gb.to(d), pr[0], pr[1], num, seg))


def test_wgrad_pairs_ragged_lists_and_empty_offsets(native_lib, oracle):
    """Lists whose lengths are no multiple of anything, offsets with no pair at all (isolated voxels),
    fewer pairs than one MFMA step."""
    from doda_amd import ops
    d = dev()
    shape, batch = [50, 50, 50], 1
    rng = np.random.default_rng(3)
    # 16 isolated voxels (only the centre offset has pairs) + a 5-voxel line
    pts = {(0, 3 * i, 3 * ((7 * i) % 16), 3 * ((5 * i) % 16)) for i in range(16)}
    pts |= {(0, 20 + i, 49, 49) for i in range(5)}
    idx = np.array(sorted(pts), dtype=np.int32)
    rng.shuffle(idx)
    pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
    m = idx.shape[0]
    xb = _bf16_round(rng.standard_normal((m, 16)).astype(np.float32))
    gb = _bf16_round(rng.standard_normal((m, 32)).astype(np.float32))
    _, ref_dw = oracle.indice_conv_backward(xb.double(), torch.zeros(3, 3, 3, 16, 32, dtype=torch.float64),
                                            gb.double(), pairs, pn, False, True)
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    pr, num, seg = ops.rulebook_pairs(tbl, m, flip=True, pad=False, with_seg=True)
    dw = ops.spconv_wgrad_pairs(xb.to(d), gb.to(d), pr[0], pr[1], num, seg)
    assert (pn == 0).sum() >= 10
    assert rel_err(dw.cpu().reshape(ref_dw.shape), ref_dw) < RTOL
    assert float(dw.cpu().reshape(27, -1)[pn == 0].abs().max()) == 0.0


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 48), (96, 112)])
def test_wgrad_bf16_strided_inverse_and_1x1_through_modules(native_lib, oracle, cin, cout, pairs, monkeypatch):
    """bf16 modules — weight gradient by the MFMA-transpose kernel over the gather table (default) or by
    the pair-list kernel (opt-in) — against the oracle in fp64 on the bf16-rounded operands: strided
    conv, its inverse, and the 1x1 convolution."""
    from doda_amd import spconv
    monkeypatch.setattr(spconv.functional, "WGRAD_PAIRS", pairs)
    d = dev()
    shape, batch = [25, 20, 23], 2
    idx = surface_voxels(cin + cout, 3000, batch, shape)
    oi, pairs, pn, oshape = oracle.indice_pairs_conv(idx, batch, shape, 2, 2, 0, 1)
    rng = np.random.default_rng(cin)
    m, mo = idx.shape[0], oi.shape[0]
    down = spconv.SparseConv3d(cin, cout, kernel_size=2, stride=2, bias=False, indice_key="d").to(d)
    up = spconv.SparseInverseConv3d(cout, cin, kernel_size=2, bias=False, indice_key="d").to(d)
    one = spconv.SubMConv3d(cin, cout, kernel_size=1, bias=False).to(d)
    # strided: a = fine x, b = coarse dy
    x = _bf16_round(rng.standard_normal((m, cin)).astype(np.float32))
    gy = _bf16_round(rng.standard_normal((mo, cout)).astype(np.float32))
    st = spconv.SparseConvTensor(x.to(d).requires_grad_(True), torch.from_numpy(idx).to(d), shape, batch)
    mid = down(st)
    mid.features.backward(gy.to(d))
    _, ref = oracle.indice_conv_backward(x.double(), down.weight.detach().cpu().double(), gy.double(), pairs, pn,
                                         False, False)
    assert rel_err(down.weight.grad.cpu(), ref) < RTOL
    # inverse: a = coarse x, b = fine dy
    xc = _bf16_round(rng.standard_normal((mo, cout)).astype(np.float32))
    gf = _bf16_round(rng.standard_normal((m, cin)).astype(np.float32))
    mid2 = spconv.SparseConvTensor(xc.to(d).requires_grad_(True), mid.indices, mid.spatial_shape, batch)
    mid2.indice_dict = mid.indice_dict
    up(mid2).features.backward(gf.to(d))
    _, ref = oracle.indice_conv_backward(xc.double(), up.weight.detach().cpu().double(), gf.double(), pairs, pn,
                                         True, False)
    assert rel_err(up.weight.grad.cpu(), ref) < RTOL
    # 1x1: dW = x^T dy, fwd and dgrad too (thin 1x1 coverage)
    g1 = _bf16_round(rng.standard_normal((m, cout)).astype(np.float32))
    st1 = spconv.SparseConvTensor(x.to(d).requires_grad_(True), torch.from_numpy(idx).to(d), shape, batch)
    y1 = one(st1)
    y1.features.backward(g1.to(d))
    w1 = one.weight.detach().cpu().double().view(cin, cout)
    wb = w1.float().to(torch.bfloat16).double()   # the kernel multiplies bf16-rounded weights
    assert rel_err(y1.features.detach().float().cpu(), x.double() @ wb) < 8e-3   # one bf16 rounding of y
    assert rel_err(st1.features.grad.float().cpu(), g1.double() @ wb.t()) < 8e-3
    assert rel_err(one.weight.grad.cpu().view(cin, cout), x.double().t() @ g1.double()) < RTOL


@pytest.mark.parametrize("cin,cout", [(32, 16), (64, 32), (224, 112)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv1x1_fwd_bwd_vs_oracle(native_lib, oracle, cin, cout, dtype):
    """1x1 SubM of the residual blocks' skip branch (reference model/unet_block.py:18-21) against the
    oracle's conv1x1 (features @ W) in fp64."""
    from doda_amd import spconv
    d = dev()
    shape, batch = [30, 30, 30], 2
    idx = surface_voxels(cin, 2500, batch, shape)
    rng = np.random.default_rng(cout)
    m = idx.shape[0]
    x = torch.from_numpy(rng.standard_normal((m, cin)).astype(np.float32)).to(dtype)
    g = torch.from_numpy(rng.standard_normal((m, cout)).astype(np.float32)).to(dtype)
    one = spconv.SubMConv3d(cin, cout, kernel_size=1, bias=False).to(d)
    st = spconv.SparseConvTensor(x.to(d).requires_grad_(True), torch.from_numpy(idx).to(d), shape, batch)
    y = one(st)
    y.features.backward(g.to(d))
    w = one.weight.detach().cpu().double().view(cin, cout)
    wk = w if dtype == torch.float32 else w.float().to(torch.bfloat16).double()
    tol = RTOL if dtype == torch.float32 else 8e-3
    assert rel_err(y.features.detach().float().cpu(), x.double() @ wk) < tol
    assert rel_err(st.features.grad.float().cpu(), g.double() @ wk.t()) < tol
    assert rel_err(one.weight.grad.cpu().view(cin, cout), x.double().t() @ g.double()) < RTOL


# ------------------------------------------------------------------ deferred weight gradients
def _need_ext():
    from doda_amd._ext import ext
    if ext is None:
        pytest.skip("compiled extension not built")
    return ext


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_wgrad_two_backward_passes_accumulate(native_lib, dtype):
    """tool/st.py:136-198 shape: two forward/backward passes before one optimizer step.  The second
    pass's deferred jobs accumulate into the first pass's gradients; result == immediate path."""
    ext = _need_ext()
    from doda_amd import spconv
    from doda_amd.spconv import functional as Fsp
    d = dev()
    shape = [30, 30, 30]
    idx = torch.from_numpy(surface_voxels(5, 4000, 1, shape)).to(d)
    torch.manual_seed(1)
    net = spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="a"),
                                  spconv.SubMConv3d(32, 16, 3, padding=1, bias=False, indice_key="a")).to(d)
    xs = [torch.randn(idx.shape[0], 16, device=d).to(dtype) for _ in range(2)]
    out = []
    try:
        for deferred in (False, True):
            assert Fsp.set_deferred_wgrad(deferred) == deferred
            net.zero_grad(set_to_none=True)
            for x in xs:
                y = net(spconv.SparseConvTensor(x, idx, shape, 1)).features.float()
                y.square().mean().backward()
            torch.cuda.synchronize()
            assert ext.pending_wgrads() == 0
            out.append([p.grad.clone() for p in net.parameters()])
    finally:
        Fsp.set_deferred_wgrad(False)
    for a, b in zip(*out):
        assert rel_err(b.cpu(), a.cpu()) < 1e-5


def test_deferred_wgrad_shared_weight_and_aborted_backward(native_lib):
    """a weight feeding two conv nodes of one graph must receive both contributions; an aborted
    backward must not leave jobs (or a latched callback flag) behind."""
    ext = _need_ext()
    from doda_amd import spconv
    from doda_amd.spconv import functional as Fsp
    d = dev()
    shape = [30, 30, 30]
    idx = torch.from_numpy(surface_voxels(6, 3000, 1, shape)).to(d)
    torch.manual_seed(2)
    conv = spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="a").to(d)
    x = torch.randn(idx.shape[0], 16, device=d)

    def loss_fn():
        st = spconv.SparseConvTensor(x, idx, shape, 1)
        return conv(conv(st)).features.square().mean()     # the same module twice

    grads = []
    try:
        for deferred in (False, True):
            Fsp.set_deferred_wgrad(deferred)
            conv.zero_grad(set_to_none=True)
            loss_fn().backward()
            torch.cuda.synchronize()
            grads.append(conv.weight.grad.clone())
        assert rel_err(grads[1].cpu(), grads[0].cpu()) < 1e-5
        # aborted backward: a hook raises after the conv node has queued its job
        conv.zero_grad(set_to_none=True)
        xr = x.clone().requires_grad_(True)
        y = conv(spconv.SparseConvTensor(xr, idx, shape, 1)).features

        def boom(g):
            raise RuntimeError("boom")
        xr.register_hook(boom)
        with pytest.raises(RuntimeError):
            y.square().mean().backward()
        # the next, healthy pass gives the right gradient and leaves nothing queued
        conv.zero_grad(set_to_none=True)
        loss_fn().backward()
        torch.cuda.synchronize()
        assert ext.pending_wgrads() == 0
        assert rel_err(conv.weight.grad.cpu(), grads[0].cpu()) < 1e-5
    finally:
        Fsp.set_deferred_wgrad(False)


def test_dsnorm_convert_and_batchnorm_checkpoint(native_lib):
    """DSNorm.convert_dsnorm, set_ds_source/target, and loading a plain BatchNorm checkpoint
    into the converted model (both domains initialised) — then the fused kernels use the right domain."""
    from doda_amd.dsnorm import DSNorm, set_ds_source, set_ds_target
    from doda_amd.model import SparseConvNet, default_cfg
    d = dev()
    cfg = default_cfg()
    src = deterministic_init(SparseConvNet(cfg), seed=3)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    net = DSNorm.convert_dsnorm(SparseConvNet(cfg))
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    bn_src = dict(src.named_modules())["unet.blocks.block0.conv_branch.0"]
    ds = dict(net.named_modules())["unet.blocks.block0.conv_branch.0"]
    assert type(ds).__name__ == "DSNorm1d"
    assert torch.equal(ds.running_mean_source, bn_src.running_mean) and torch.equal(ds.running_var_target, bn_src.running_var)
    net.to(d).train()
    net.apply(set_ds_target)
    assert ds.domain_label == 1
    net.apply(set_ds_source)
    assert ds.domain_label == 0


# ------------------------------------------------------------------ BatchNorm statistics in the conv epilogues
def _ext_or_skip():
    from doda_amd._ext import ext
    if ext is None:
        pytest.skip("compiled extension not built")
    return ext


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,cin,cout", [(60000, 16, 16), (9000, 32, 48), (5000, 64, 112), (12000, 4, 16)])
def test_conv_epilogue_statistics_forward_and_backward(native_lib, dtype, m, cin, cout):
    """doda_spconv_gather_ex: (sum y, sum y^2) of the stored output, and for a data-grad call the
    BatchNorm-backward sums (sum dz, sum dz*xhat) with the ReLU mask recomputed from bn_x."""
    ext = _ext_or_skip()
    from doda_amd import ops
    d = dev()
    shape = [60, 50, 40]
    idx = torch.from_numpy(surface_voxels(m % 97, m, 2, shape)).to(d)
    n = idx.shape[0]
    tbl = ops.rulebook_subm(idx, shape, 2, 3)
    torch.manual_seed(cin + cout)
    x = torch.randn(n, cin, device=d).to(dtype)
    w = torch.randn(27, cin, cout, device=d) * 0.1
    res = torch.randn(n, cout, device=d).to(dtype)
    y_plain = ops.spconv_gather(x, w, tbl, n, 0, cout, residual=res)
    wt = torch.nn.Parameter(w.view(3, 3, 3, cin, cout).clone())
    y, stats = ext.indice_conv_stats(x, wt, tbl, tbl, n, 2, None, None, res)
    assert torch.equal(y.detach(), y_plain)
    from tests.util import stats_sums
    assert stats is not None and stats.shape[1] == 2 and tuple(stats_sums(stats).shape) == (2, cout)
    yf = y.detach().double()
    s1, s2 = stats_sums(stats)
    assert rel_err(s1.cpu(), yf.sum(0).cpu()) < 1e-5
    assert rel_err(s2.cpu(), (yf * yf).sum(0).cpu()) < 1e-5
    # backward sums through the public gather_ex path: use the ext's conv node with a BatchNorm in front
    from doda_amd import spconv
    torch.manual_seed(3)
    bn = torch.nn.BatchNorm1d(cin, eps=1e-4).to(d).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    if cin % 4:
        return
    seq = spconv.SparseSequential(bn, torch.nn.ReLU(), spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(d))
    seq.train()
    outs = []
    from doda_amd.spconv import functional as Fsp
    try:
        for fused in (False, True):
            Fsp.set_bn_fusion(fused)
            bn.zero_grad(set_to_none=True)
            seq[2].zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            out = seq(spconv.SparseConvTensor(xin, idx, shape, 2)).features
            out.float().square().mean().backward()
            outs.append((out.detach().clone(), xin.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    finally:
        Fsp.set_bn_fusion(True)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(*outs):
        assert rel_err(b.float().cpu(), a.float().cpu()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_step_with_and_without_bn_fusion(native_lib, dtype):
    """Whole U-Net training step: loss, every parameter gradient and the BatchNorm running statistics agree
    between the fused path (statistics in the conv epilogues, forward and backward) and the standalone
    BatchNorm kernels."""
    _ext_or_skip()
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch
    from doda_amd.spconv import functional as Fsp
    d = dev()
    cfg = default_cfg()
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in make_batch(2, 30000, 5).items()}
    net = deterministic_init(SparseConvNet(cfg), seed=1).to(d).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    runs = []
    try:
        for fused in (False, True):
            Fsp.set_bn_fusion(fused)
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=dtype), bd["labels"])
            loss.backward()
            torch.cuda.synchronize()
            runs.append((float(loss), {k: p.grad.double().cpu() for k, p in net.named_parameters()},
                         {k: v.double().cpu() for k, v in net.state_dict().items() if "running" in k}))
    finally:
        Fsp.set_bn_fusion(True)
    (l0, g0, r0), (l1, g1, r1) = runs
    tol_l, tol_g = (1e-5, 2e-3) if dtype == torch.float32 else (3e-3, 8e-2)
    assert abs(l0 - l1) <= tol_l * abs(l0), (l0, l1)
    top = max(float(g.norm()) for g in g0.values())
    for k in g0:
        na, nb = float(g0[k].norm()), float(g1[k].norm())
        assert np.isfinite(nb)
        # bf16: gradients two orders of magnitude below the largest are rounding noise at the deep levels
        # (a few dozen rows per BatchNorm, 1/sqrt(var + 1e-4) up to 100); fp32 checks every parameter
        if dtype == torch.float32 or na >= 1e-2 * top:
            assert abs(na - nb) <= tol_g * na + 1e-9, (k, na, nb)
    for k in r0:
        assert rel_err(r1[k], r0[k]) < (1e-5 if dtype == torch.float32 else 2e-2), k


# ------------------------------------------------------------------ thin spots of round 1 (VERDICT "tighten")
def test_maxpool_backward_elementwise_vs_oracle(native_lib, oracle):
    """Indice max-pool backward against the oracle's restatement of spconv's maxPoolBwd, element by element,
    with ties (duplicated maxima receive the gradient once each)."""
    from doda_amd import spconv
    from tests.util import random_voxels
    shape, batch = [12, 10, 8], 2
    idx = random_voxels(9, 500, batch, shape)
    rng = np.random.default_rng(4)
    x = rng.integers(-3, 4, size=(idx.shape[0], 8)).astype(np.float32)   # small integers: many exact ties
    oi, pairs, pn, _ = oracle.indice_pairs_conv(idx, batch, shape, 2, 2, 0, 1)
    y = oracle.indice_maxpool(torch.from_numpy(x), pairs, pn, oi.shape[0])
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    ref = oracle.indice_maxpool_backward(torch.from_numpy(x), y, torch.from_numpy(dy), pairs, pn)
    pool = spconv.SparseMaxPool3d(2, 2)
    xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
    out = pool(spconv.SparseConvTensor(xt, torch.from_numpy(idx).to(dev()), shape, batch))
    assert np.array_equal(out.features.detach().cpu().numpy(), y.numpy())
    out.features.backward(torch.from_numpy(dy).to(dev()))
    assert np.array_equal(xt.grad.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_tail_layer_properties(native_lib, dtype):
    """BASELINE config 2 at full size: 4 x ~150k voxels, the widest level-1 layer shape x 6 (192 -> 96
    channels: 231 MB bf16 / 463 MB fp32 of input features — the regime of the 32-bit buffer offsets and
    the < 2 GB guards).  Size-independent checks: a centre-only identity kernel copies its channels
    exactly; linearity; sampled rows against a direct fp64 evaluation of y[t] = sum_o x[nbr[o][t]] W[o];
    the weight gradient of sampled (offset, ci, co) entries against fp64."""
    from doda_amd import ops
    from doda_amd.scene import make_batch
    d = dev()
    b = make_batch(4, 150000, 1000)
    idx = b["voxel_locs"].int().to(d)
    m = idx.shape[0]
    assert m > 590000
    tbl = ops.rulebook_subm(idx, b["spatial_shape"], 4, 3)
    cin, cout = 192, 96
    g = torch.Generator(device="cpu").manual_seed(0)
    x1 = torch.randn(m, cin, generator=g).to(d).to(dtype)
    x2 = torch.randn(m, cin, generator=g).to(d).to(dtype)
    w = (torch.randn(27, cin, cout, generator=g) * 0.05).to(d)
    # identity on the centre offset
    wi = torch.zeros(27, cin, cout, device=d)
    wi[13, :cout, :] = torch.eye(cout, device=d)
    yi = ops.spconv_gather(x1, wi, tbl, m, 0, cout)
    import os
    if dtype == torch.float32 and int(os.environ.get("DODA_F32_SPLIT_ROWS", "-1")) >= 0:
        # (opt-in bf16 head + tail splits of both fp32 operands — x = hi + lo + e with |e| <= 2^-17 |x| —: the identity
        # kernel then copies to 2^-16 relative, per element, instead of bit for bit)
        assert float(((yi - x1[:, :cout]).abs() / x1[:, :cout].abs().clamp_min(1e-30)).max()) <= 2.0 ** -16
    else:
        assert torch.equal(yi, x1[:, :cout].contiguous())
    y1 = ops.spconv_gather(x1, w, tbl, m, 0, cout).float()
    y2 = ops.spconv_gather(x2, w, tbl, m, 0, cout).float()
    if dtype == torch.float32:   # linearity (exact inputs; bf16 would round x1 + x2)
        y12 = ops.spconv_gather(x1 + 0.5 * x2, w, tbl, m, 0, cout)
        assert float((y12 - (y1 + 0.5 * y2)).abs().max()) < 2e-4 * float(y1.abs().max())
    # sampled rows, fp64 (rows near the end of the buffers included)
    rows = torch.cat([torch.randint(0, m, (1500,), generator=g), torch.arange(m - 300, m)]).to(d)
    wk = w if dtype == torch.float32 else w.to(torch.bfloat16).float()
    nb = tbl[:, rows].long()                                  # [27, R]
    xs = torch.where((nb >= 0).unsqueeze(-1), x1[nb.clamp_min(0)].double(), torch.zeros((), dtype=torch.float64, device=d))
    ref = torch.einsum("orc,ocd->rd", xs, wk.double())
    tol = RTOL if dtype == torch.float32 else 8e-3
    assert rel_err(y1[rows].cpu(), ref.cpu()) < tol
    # weight gradient: all offsets, a sampled block of channels
    gy = torch.randn(m, cout, generator=g).to(d).to(dtype)
    dw = ops.spconv_wgrad(x1, gy, tbl, m)
    ci, co = slice(40, 56), slice(80, 96)
    ref_dw = torch.empty(27, 16, 16, dtype=torch.float64, device=d)
    for o in range(27):
        nbo = tbl[o].long()
        ok = nbo >= 0
        ref_dw[o] = x1[nbo[ok]][:, ci].double().t() @ gy[ok][:, co].double()
    assert rel_err(dw[:, ci, co].cpu(), ref_dw.cpu()) < (RTOL if dtype == torch.float32 else 2e-3)


def test_one_cm_scene_rulebooks_and_step(native_lib, oracle):
    """BASELINE config 5 shape: 1 cm voxels, ~500 k active voxels in one scene.  Rulebooks bit-exact
    against the oracle at that size, and the U-Net training step on the tile kernels equals the step on the dense-table
    kernels (loss 2 %, every parameter gradient within 10 % in norm: two bf16 evaluation orders)."""
    from doda_amd import ops
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch
    d = dev()
    b = make_batch(1, 500000, 77, voxel_scale=100)
    idx = b["voxel_locs"].int()
    shape = [int(v) for v in b["spatial_shape"]]
    assert idx.shape[0] > 450000
    pairs, pn = oracle.indice_pairs_subm(idx.numpy(), 1, shape, 3)
    tbl = ops.rulebook_subm(idx.to(d), shape, 1, 3)
    got, num = ops.rulebook_pairs(tbl, idx.shape[0], flip=True)
    assert np.array_equal(num.cpu().numpy(), pn) and np.array_equal(got.cpu().numpy(), pairs)
    oi, dpairs, dpn, oshape = oracle.indice_pairs_conv(idx.numpy(), 1, shape, 2, 2, 0, 1)
    out_idx, child, par_off, out_shape = ops.rulebook_down2(idx.to(d), shape, 1)
    assert out_shape == oshape and np.array_equal(out_idx.cpu().numpy(), oi)
    gp, gn = ops.rulebook_pairs(par_off, idx.shape[0], flip=False)
    assert np.array_equal(gn.cpu().numpy(), dpn) and np.array_equal(gp.cpu().numpy(), dpairs)
    # the training step on that scene: tile kernels (tilebooks) against the dense-table gather kernels, bf16 — loss and every
    # parameter gradient within the tolerances of the 2 cm comparison (tests/test_gpu_round4.py, config 2)
    from doda_amd import spconv
    cfg = default_cfg()
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
    got = {}
    old_tile = spconv.ops.TILE_KERNEL
    try:
        for tiled in (True, False):
            spconv.ops.TILE_KERNEL = tiled
            net = deterministic_init(SparseConvNet(cfg), seed=2).to(d).train()
            loss = cross_entropy(voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.bfloat16), bd["labels"])
            loss.backward()
            torch.cuda.synchronize()
            assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in net.parameters())
            got[tiled] = (float(loss.detach()), {k: p.grad.detach().float().clone() for k, p in net.named_parameters()})
            del net, loss
    finally:
        spconv.ops.TILE_KERNEL = old_tile
    (l1, g1), (l0, g0) = got[True], got[False]
    assert abs(l0 - l1) < 2e-2 * abs(l0), (l0, l1)
    for k, a in g0.items():
        assert (a - g1[k]).norm().item() <= 0.1 * a.norm().item() + 1e-6, k


def _huge_grid_scene(seed, batch, shape, n):
    """n distinct voxels in clusters spread over a grid of more than 2^32 cells, some in the last cells
    of the last sample (largest cell ids), with neighbours inside each cluster."""
    rng = np.random.RandomState(seed)
    centres = np.stack([rng.randint(0, batch, 64)] + [rng.randint(4, s - 4, 64) for s in shape], 1)
    centres[0] = [batch - 1] + [s - 3 for s in shape]
    centres[1] = [0, 2, 2, 2]
    pick = centres[rng.randint(0, 64, 2 * n)]
    pts = pick + np.concatenate([np.zeros((2 * n, 1), np.int64), rng.randint(-2, 3, (2 * n, 3))], 1)
    pts[:, 1:] = np.clip(pts[:, 1:], 0, np.array(shape) - 1)
    pts = np.unique(pts, axis=0)
    pts = pts[rng.permutation(len(pts))][:n]
    return np.ascontiguousarray(pts, dtype=np.int32)


def test_rulebooks_on_a_grid_of_more_than_2_pow_32_cells(native_lib, oracle):
    """batch x X x Y x Z >= 2^32 (batch 4 of 1 cm scenes, 2000 x 2000 x 600 each = 9.6e9 cells): the hash
    words give the cell id more than 32 bits and the row number fewer (common.hpp HashFmt); all three
    rulebook builders stay bit-exact against the oracle, which keys on 64-bit cell ids.  A grid whose
    cell id and row number cannot share 64 bits is still refused with DODA_ERR_GRID_TOO_LARGE."""
    from doda_amd import ops
    from doda_amd._lib import DodaNativeError
    d = dev()
    batch, shape = 4, [2000, 2000, 600]
    idx_h = _huge_grid_scene(5, batch, shape, 3000)
    cell = ((idx_h[:, 0].astype(np.int64) * shape[0] + idx_h[:, 1]) * shape[1] + idx_h[:, 2]) * shape[2] + idx_h[:, 3]
    assert cell.max() >= 2 ** 33 and len(np.unique(cell)) == len(cell)
    idx = torch.from_numpy(idx_h).to(d)
    m = idx.shape[0]
    pairs, pn = oracle.indice_pairs_subm(idx_h, batch, shape, 3)
    assert pn.sum() > 4 * m                                        # the clusters do have neighbours
    got, num = ops.rulebook_pairs(ops.rulebook_subm(idx, shape, batch, 3), m, flip=True)
    assert np.array_equal(num.cpu().numpy(), pn) and np.array_equal(got.cpu().numpy(), pairs)
    oi, dpairs, dpn, oshape = oracle.indice_pairs_conv(idx_h, batch, shape, 2, 2, 0, 1)
    out_idx, child, par_off, out_shape = ops.rulebook_down2(idx, shape, batch)
    assert out_shape == oshape and np.array_equal(out_idx.cpu().numpy(), oi)
    gp, gn = ops.rulebook_pairs(par_off, m, flip=False)
    assert np.array_equal(gn.cpu().numpy(), dpn) and np.array_equal(gp.cpu().numpy(), dpairs)
    # generic builder (3x3x3 stride 1 padding 1: the output grid is as large as the input grid; the
    # hash values are row * 27 + rank)
    oi, cpairs, cpn, oshape = oracle.indice_pairs_conv(idx_h, batch, shape, 3, 1, 1, 1)
    out_idx, tbl, tbl_rev, out_shape = ops.rulebook_conv(idx, shape, batch, 3, 1, 1, 1)
    assert out_shape == oshape and np.array_equal(out_idx.cpu().numpy(), oi)
    gp, gn = ops.rulebook_pairs(tbl_rev, m, flip=False)
    assert np.array_equal(gn.cpu().numpy(), cpn) and np.array_equal(gp.cpu().numpy(), cpairs)
    # the limit that remains: bits(cell id) + bits(row number) <= 64
    few = torch.tensor([[0, 1, 2, 3], [1, 5, 6, 7]], dtype=torch.int32, device=d)
    ops.rulebook_subm(few, [2048, 2048, 512], 2, 3)                # 2^32 cells: refused before, fine now
    ops.rulebook_subm(few, [1 << 20, 1 << 20, 1 << 20], 4, 3)      # 2^62 cells, 2 rows
    with pytest.raises(DodaNativeError, match="-3|grid|GRID"):
        ops.rulebook_subm(few, [1 << 21, 1 << 21, 1 << 21], 2, 3)  # 2^64 cells


def test_python_glue_without_the_compiled_extension(native_lib):
    """DODA_NO_EXT=1: the pure Python / ctypes glue drives the same kernels; the reference-model golden
    must hold on that route too (round 1 never exercised it on the GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DODA_NO_EXT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_unet.py", "-k",
                        "golden or bf16_tracks"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("momentum,dampening,nesterov,wd", [(0.9, 0.0, False, 1e-4), (0.9, 0.0, True, 1e-4),
                                                            (0.8, 0.1, False, 0.0), (0.0, 0.0, False, 5e-4)])
def test_fused_sgd_is_torch_sgd(native_lib, momentum, dampening, nesterov, wd):
    """doda_amd.optim.FusedSGD (one launch for all tensors) against torch.optim.SGD: bit-identical to torch's
    fused multi-tensor path, within an ulp of its single-tensor path (float instead of double products);
    odd sizes, unaligned views, a parameter without gradient, lr changed between steps (the reference's
    schedulers write param_group["lr"]), state_dict round trip."""
    from doda_amd.optim import FusedSGD
    d = dev()
    g = torch.Generator().manual_seed(3)
    base = torch.randn(40000, generator=g)
    shapes = [(27, 16, 16), (16,), (1,), (3, 3, 3, 3, 16), (224,), (20, 16), (1031,), (7, 5)]

    def make():
        ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(d)) for s in shapes]
        ps.append(torch.nn.Parameter(base.to(d)[1:1 + 4097].clone()))
        ps.append(torch.nn.Parameter(torch.zeros(5, device=d)))   # never gets a gradient
        return ps
    g = torch.Generator().manual_seed(3); pa = make()
    g = torch.Generator().manual_seed(3); pb = make()
    g = torch.Generator().manual_seed(3); pc = make()
    kw = dict(lr=0.05, momentum=momentum, dampening=dampening, nesterov=nesterov, weight_decay=wd)
    oa, ob, oc = FusedSGD(pa, **kw), torch.optim.SGD(pb, fused=True, **kw), torch.optim.SGD(pc, foreach=False, **kw)
    gg = torch.Generator().manual_seed(11)
    for it in range(4):
        if it == 2:
            for o in (oa, ob, oc):
                o.param_groups[0]["lr"] = 0.0125
        if it == 3:   # checkpoint round trip of the optimizer state into a fresh FusedSGD
            sd = oa.state_dict()
            oa = FusedSGD(pa, **kw)
            oa.load_state_dict(sd)
        grads = [torch.randn(p.shape, generator=gg).to(d) for p in pa[:-1]]
        for ps in (pa, pb, pc):
            for p, gr in zip(ps[:-1], grads):
                p.grad = gr.clone()
        oa.step(); ob.step(); oc.step()
    torch.cuda.synchronize()
    for a, b, c in zip(pa, pb, pc):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, rtol=2e-6, atol=1e-7)
    if momentum:
        for a, b in zip(pa[:-1], pb[:-1]):
            assert torch.equal(oa.state[a]["momentum_buffer"], ob.state[b]["momentum_buffer"])
    assert set(oa.state_dict()["state"].keys()) == set(ob.state_dict()["state"].keys())
