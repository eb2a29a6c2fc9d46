"""Round-4 GPU tests: the default kernels in the regime the bench runs them in.

conv_tile launches min(round8(tiles), 768) persistent workgroups (512 for 64-byte rows) and wgrad_dma16
min(round8(tiles), 256); the step selects wgrad_dma16 only for rulebooks of >= 262 144 rows.  Every earlier
kernel-level parity test stayed at <= 157 tiles, i.e. ONE tile per workgroup: the next-tile list prefetch, the
cross-tile statistics accumulators and the two-buffer DMA-ahead pipeline with accumulators carried across tiles
never ran under an oracle comparison.  Here: >= 300 k rows (>= 1172 tiles: >= 2 tiles per conv_tile workgroup,
>= 4 per wgrad_dma16 workgroup), a ragged last tile, and a block of shuffled rows whose tiles reference more distinct
rows than a tilebook lists (tiles "without a list", served from the dense table inside the kernels) — all against
oracle.indice_conv / indice_conv_backward on ORACLE-built pair lists (reference semantics:
model/unet_block.py:26,29; cfgs/scannet/spconv.yaml:27), plus one BASELINE-config-2-size step (4 x 150 k voxels).
"""
import numpy as np
import pytest
import torch

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


_SCENE = {}


def _big_scene(oracle):
    """Two synthetic ScanNet-shaped scenes (doda_amd.scene, the bench's generator) of ~160 k voxels each in their
    natural first-touch order, with one block of 4096 consecutive rows shuffled among themselves: the 16 tiles of that
    block reference ~1800 distinct rows each (> TB_UMAX = 1024) and lose their lists.  Returns (idx int32 [n,4], shape,
    batch, oracle pairs, oracle pair counts); cached for the module (the oracle rulebook takes ~3 s)."""
    if "v" not in _SCENE:
        from doda_amd.scene import make_batch
        b = make_batch(2, 160000, 4242)
        idx = b["voxel_locs"].int().numpy().copy()
        shape = [int(s) for s in b["spatial_shape"]]
        n = idx.shape[0]
        assert n >= 300000 and n % 256 != 0, n                    # >= 1172 tiles, ragged last tile
        lo = (n // 3) // 256 * 256 + 128                          # (the block straddles tile boundaries)
        perm = np.random.default_rng(5).permutation(4096)
        idx[lo:lo + 4096] = idx[lo:lo + 4096][perm]
        idx = np.ascontiguousarray(idx)
        pairs, pn = oracle.indice_pairs_subm(idx, 2, shape, 3)
        _SCENE["v"] = (idx, shape, 2, pairs, pn)
    return _SCENE["v"]


def _hip_rulebook(idx, shape, batch, pairs, pn):
    from doda_amd import ops
    d = dev()
    n = idx.shape[0]
    tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    hip_pairs, hip_pn = ops.rulebook_pairs(tbl, n, flip=True)
    assert np.array_equal(hip_pn.cpu().numpy(), pn) and np.array_equal(hip_pairs.cpu().numpy(), pairs)
    tb = ops.tilebook_build(tbl)
    assert tb is not None
    n_over = tb[-8:].view(torch.int32).cpu().tolist()             # tiles above 960 / above 1024 distinct rows
    assert n_over[1] >= 8, n_over                                  # tiles WITHOUT a list are in the launch
    nt = (n + 255) // 256
    assert nt >= 1172 and n_over[1] < nt // 20
    return tbl, tb


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (32, 16), (16, 32)])
def test_tile_kernels_multi_tile_regime_vs_oracle(native_lib, oracle, cin, cout):
    """conv_tile MODE 0 (32-byte rows) and MODE 1 (64-byte rows), forward (layout 0) and data gradient (layout 2), at
    >= 2 tiles per persistent workgroup: plain, with the fused residual, and with the statistics epilogue (the sum of
    the per-workgroup rows against the fp64 column sums of the ORACLE's output).  Operands bf16-representable: every
    product is exact in fp32, only the summation order differs — 1e-4 on fp32 outputs (north_star), one bf16 rounding
    step on bf16 outputs."""
    from doda_amd import ops
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    n = idx.shape[0]
    tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    g = torch.Generator().manual_seed(1000 * cin + cout)
    x = torch.randn(n, cin, generator=g).bfloat16()
    dy = torch.randn(n, cout, generator=g).bfloat16()
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.1).bfloat16().float()
    ref_y = oracle.indice_conv(x.double(), w.double(), pairs, pn, n, False, True)
    ref_dx, _ = oracle.indice_conv_backward(x.double(), w.double(), dy.double(), pairs, pn, False, True)
    xd, dyd, wd = x.to(d), dy.to(d), w.to(d).view(27, cin, cout)

    for inp, layout, nc, ref in ((xd, 0, cout, ref_y), (dyd, 2, cin, ref_dx)):
        scale = float(ref.abs().max())
        y32 = ops.spconv_gather(inp, wd, tbl, n, layout, nc, out_f32=True, tilebook=tb)
        assert rel_err(y32.cpu(), ref) < 1e-4, (layout, "fp32 out")
        yb = ops.spconv_gather(inp, wd, tbl, n, layout, nc, tilebook=tb)
        assert yb.dtype == torch.bfloat16
        tol = 2.0 ** -7 * ref.abs() + 1e-5 * scale                # the fp64 result rounded once (+ last-bit noise)
        assert bool(((yb.float().cpu().double() - ref).abs() <= tol).all()), (layout, "bf16 out")
        # residual + statistics epilogue, the instantiation the step launches
        res = torch.randn(n, nc, generator=g).bfloat16()
        yr, st = ops.spconv_gather(inp, wd, tbl, n, layout, nc, tilebook=tb, residual=res.to(d), want_stats=True)
        want = ref + res.double()
        tol = 2.0 ** -7 * want.abs() + 1e-5 * float(want.abs().max())
        assert bool(((yr.float().cpu().double() - want).abs() <= tol).all()), (layout, "residual")
        assert 2 <= st.shape[0] <= 768                             # one row per persistent workgroup
        # contract (include/doda_hip.h, doda_conv_epilogue.stats): sums over y AS STORED (conv + residual, after the
        # bf16 rounding), accumulated across all tiles of a workgroup: against fp64 column sums of the stored tensor,
        # and against the oracle's values (which differ from the stored ones by one rounding per element)
        tot = st.double().sum(0).cpu()
        yf = yr.double().cpu()
        assert rel_err(tot[0], yf.sum(0)) < 1e-5 and rel_err(tot[1], (yf * yf).sum(0)) < 1e-5, layout
        assert rel_err(tot[1], (want * want).sum(0)) < 2e-3 and rel_err(tot[0], want.sum(0)) < 2e-2, layout
        ys, st2 = ops.spconv_gather(inp, wd, tbl, n, layout, nc, tilebook=tb, want_stats=True)
        tot2, yf2 = st2.double().sum(0).cpu(), ys.double().cpu()
        assert rel_err(tot2[0], yf2.sum(0)) < 1e-5 and rel_err(tot2[1], (yf2 * yf2).sum(0)) < 1e-5, layout
        assert rel_err(tot2[1], (ref * ref).sum(0)) < 2e-3, layout
        # repeatable bit for bit (fixed-order reductions; no atomics on floats)
        yr2, st3 = ops.spconv_gather(inp, wd, tbl, n, layout, nc, tilebook=tb, residual=res.to(d), want_stats=True)
        assert torch.equal(yr, yr2) and torch.equal(st, st3)


def test_wgrad_dma16_in_the_regime_the_step_selects_it(native_lib, oracle):
    """wgrad_dma16 the way the step passes its jobs — table + tilebook + PAIR LISTS present, so that the >= 262 144-row
    gate of classify() (csrc/spconv_wgrad.hip) is what picks the kernel —, four layers of one rulebook in one call, one
    of them accumulating into an existing gradient: against oracle.indice_conv_backward on the oracle's pair lists.
    That the LDS-staged kernel (and not the pair-list kernel) ran: bit-equal to the same call without pair lists (which
    can only take the tile kernel), and NOT bit-equal to the call without a tilebook."""
    from doda_amd import ops
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    n = idx.shape[0]
    assert n >= 262144
    tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    pr, num, seg = ops.rulebook_pairs(tbl, n, flip=True, pad=False, with_seg=True)
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn(n, 16, generator=g).bfloat16() for _ in range(4)]
    dys = [torch.randn(n, 16, generator=g).bfloat16() for _ in range(4)]
    w0 = torch.zeros(3, 3, 3, 16, 16, dtype=torch.float64)
    refs = [oracle.indice_conv_backward(x.double(), w0, dy.double(), pairs, pn, False, True)[1] for x, dy in zip(xs, dys)]
    base = torch.randn(27, 16, 16, generator=g)
    plist = (pr[0], pr[1], num, seg)

    def jobs(with_pairs, with_tb, acc):
        out = []
        for k in range(4):
            out.append((xs[k].to(d), dys[k].to(d), tbl, n, plist if with_pairs else None,
                        acc if k == 2 else None, tb if with_tb else None))
        return out
    acc = base.clone().to(d)
    got = ops.spconv_wgrad_multi(jobs(True, True, acc))
    for k in range(4):
        val = got[k].cpu() - (base if k == 2 else 0)
        assert rel_err(val.reshape(refs[k].shape), refs[k]) < 1e-4, k
    acc2 = base.clone().to(d)
    tile_only = ops.spconv_wgrad_multi(jobs(False, True, acc2))
    assert all(torch.equal(a, b) for a, b in zip(got, tile_only))
    acc3 = base.clone().to(d)
    pairs_only = ops.spconv_wgrad_multi(jobs(True, False, acc3))
    assert not all(torch.equal(a, b) for a, b in zip(got, pairs_only))     # another kernel, another summation order
    for k in range(4):
        val = pairs_only[k].cpu() - (base if k == 2 else 0)
        assert rel_err(val.reshape(refs[k].shape), refs[k]) < 1e-4, k


@pytest.mark.parametrize("ca,cb,big", [(32, 16, True), (16, 32, True), (32, 32, False), (32, 16, False), (64, 32, False)])
def test_wgrad_over_the_tilebook_for_32_channel_sides(native_lib, oracle, ca, cb, big):
    """Round 4: layers with 32 channels on either side run the LDS-staged weight gradient as 16 x 16 channel blocks over
    row-strided halves of x / dy (doda_wdma::Block) — the level-1 32 -> 16 layer of the U-Net (model/unet_block.py:83-88,
    blocks_tail) no longer needs the rulebook's pair lists.  Against oracle.indice_conv_backward on oracle-built pairs, at
    the row count where the step selects the kernel (big) and on a small ragged scene; with an accumulate job; and
    bit-equal on a second call (fixed-order reduce)."""
    from doda_amd import ops
    d = dev()
    if big:
        idx, shape, batch, pairs, pn = _big_scene(oracle)
        tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    else:
        from tests.util import surface_voxels
        shape, batch = [40, 36, 30], 1
        idx = surface_voxels(3001, 3001, batch, shape)
        key = ((idx[:, 0].astype(np.int64) * shape[0] + idx[:, 1]) * shape[1] + idx[:, 2]) * shape[2] + idx[:, 3]
        idx = np.ascontiguousarray(idx[np.argsort(key, kind="stable")].astype(np.int32))
        pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
        tbl = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
        tb = ops.tilebook_build(tbl)
    n = idx.shape[0]
    g = torch.Generator().manual_seed(100 * ca + cb)
    x = torch.randn(n, ca, generator=g).bfloat16()
    dy = torch.randn(n, cb, generator=g).bfloat16()
    ref = oracle.indice_conv_backward(x.double(), torch.zeros(3, 3, 3, ca, cb, dtype=torch.float64), dy.double(), pairs, pn,
                                      False, True)[1]
    base = torch.randn(27, ca, cb, generator=g)
    acc = base.clone().to(d)
    xd, dyd = x.to(d), dy.to(d)
    got = ops.spconv_wgrad_multi([(xd, dyd, tbl, n, None, None, tb), (xd, dyd, tbl, n, None, acc, tb)])
    assert rel_err(got[0].cpu().reshape(ref.shape), ref) < 1e-4
    assert rel_err((got[1].cpu() - base).reshape(ref.shape), ref) < 1e-4
    # bit-equal when the same call is repeated (fixed-order reduce; the workgroups' chunks — hence the summation order —
    # depend on how many channel blocks a launch holds, so a call with other jobs agrees to rounding only)
    acc2 = base.clone().to(d)
    again = ops.spconv_wgrad_multi([(xd, dyd, tbl, n, None, None, tb), (xd, dyd, tbl, n, None, acc2, tb)])
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])
    alone = ops.spconv_wgrad_multi([(xd, dyd, tbl, n, None, None, tb)])[0]
    assert rel_err(alone.cpu(), got[0].cpu()) < 1e-5
    dense = ops.spconv_wgrad_multi([(xd, dyd, tbl, n)])[0]              # the gather-table kernel: another summation order
    assert rel_err(dense.cpu().reshape(ref.shape), ref) < 1e-4 and not torch.equal(dense, got[0])


def _config2_step(dtype, tiled, batch_dev, n_steps=1):
    """bench.py's step (deferred weight gradients -> doda_spconv_wgrad_multi, FusedSGD left out: one step's loss and
    gradients) on a resident batch, tilebooks + wgrad_dma16 on or off."""
    from doda_amd import spconv
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.spconv import functional as Fsp
    from tests.util import deterministic_init
    d = dev()
    cfg = default_cfg()
    old = spconv.ops.TILE_KERNEL
    assert Fsp.set_deferred_wgrad(True)
    try:
        spconv.ops.TILE_KERNEL = tiled
        net = deterministic_init(SparseConvNet(cfg), seed=3).to(d).train()
        net.zero_grad(set_to_none=True)
        loss = cross_entropy(voxelize_and_run(cfg, net, batch_dev, d, feature_dtype=dtype), batch_dev["labels"],
                             ignore_index=255)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        assert all(v is not None for v in grads.values())
        return float(loss.detach()), grads
    finally:
        spconv.ops.TILE_KERNEL = old
        Fsp.set_deferred_wgrad(False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_config2_size_step_tile_path_vs_dense_table_path(native_lib, dtype, monkeypatch):
    """BASELINE config 2 at full size (4 scenes x ~150 k voxels, cfgs/scannet/spconv.yaml:16-27): the step with
    tilebooks (conv_tile at 3+ tiles per workgroup, wgrad_dma16 selected by the row gate) against the same step on the
    dense gather tables — loss and every parameter gradient within the tolerances of
    tests/test_gpu_tile.py::test_unet_step_with_and_without_tilebooks."""
    _ext_or_skip()
    from doda_amd.scene import make_batch
    import doda_amd.model as dmodel
    d = dev()
    batch = make_batch(4, 150000, 1000)
    bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert bd["voxel_locs"].shape[0] >= 4 * 140000
    if dtype == torch.float32 and dmodel.tile_levels_for(dtype) == 0:
        monkeypatch.setattr(dmodel, "tile_levels_for", lambda dt: 1)          # fp32 tile path: opt in
    l0, g0 = _config2_step(dtype, False, bd)
    l1, g1 = _config2_step(dtype, True, bd)
    assert np.isfinite(l0) and np.isfinite(l1)
    tol = (2e-2, 0.1) if dtype == torch.bfloat16 else (1e-4, 2e-2)
    assert abs(l0 - l1) < tol[0] * abs(l0), (l0, l1)
    for k, a in g0.items():
        b = g1[k]
        assert (a.float() - b.float()).norm().item() <= tol[1] * a.float().norm().item() + 1e-6, k


@pytest.mark.parametrize("kind,m,shape,batch", [("surface", 30000, [80, 70, 60], 2), ("random", 5000, [33, 21, 47], 3),
                                                 ("bench", 150000, None, 1)])
def test_rulebook_direct_address_grid_equals_the_hash_build(native_lib, oracle, kind, m, shape, batch):
    """Round 4: doda_rulebook_subm / doda_rulebook_down2_assign with room for a direct-address grid in the workspace
    (include/doda_hip.h) against the same calls on the minimum workspace (64-bit hash table): SubM tables, k2 s2 parents /
    offsets / first-touch output rows / count bit-equal, and the SubM table equal to the oracle's pair lists."""
    import ctypes as C
    from doda_amd import ops
    from doda_amd._lib import lib, check
    from tests.util import random_voxels, surface_voxels
    d = dev()
    if kind == "bench":
        from doda_amd.scene import make_batch
        b = make_batch(1, m, 77)
        idx, shape, batch = b["voxel_locs"].int().numpy(), [int(v) for v in b["spatial_shape"]], 1
    else:
        idx = (surface_voxels if kind == "surface" else random_voxels)(5, m, batch, shape)
    n = idx.shape[0]
    ind = torch.from_numpy(np.ascontiguousarray(idx)).to(d)
    shape_c = (C.c_int32 * 3)(*shape)
    stream = torch.cuda.current_stream().cuda_stream
    base = lib().doda_rulebook_workspace_bytes(n)

    def build(with_grid, coarse):
        shp = [(v - 2) // 2 + 1 for v in shape] if coarse else shape
        cells = batch * shp[0] * shp[1] * shp[2]
        nbytes = (base + 255) // 256 * 256 + 4 * cells if with_grid else base
        return torch.empty(nbytes, dtype=torch.uint8, device=d)

    tables = []
    for with_grid in (False, True):
        ws = build(with_grid, False)
        nbr = torch.empty((27, n), dtype=torch.int32, device=d)
        check(lib().doda_rulebook_subm(ind.data_ptr(), n, shape_c, batch, 3, nbr.data_ptr(), n, ws.data_ptr(), ws.numel(), stream),
              "doda_rulebook_subm")
        ws2 = build(with_grid, True)
        parent = torch.empty(n, dtype=torch.int32, device=d)
        off = torch.empty(n, dtype=torch.int32, device=d)
        out_idx = torch.zeros((n, 4), dtype=torch.int32, device=d)
        count = torch.zeros(1, dtype=torch.int32, device=d)
        check(lib().doda_rulebook_down2_assign(ind.data_ptr(), n, shape_c, batch, parent.data_ptr(), off.data_ptr(),
                                               out_idx.data_ptr(), count.data_ptr(), ws2.data_ptr(), ws2.numel(), stream),
              "doda_rulebook_down2_assign")
        torch.cuda.synchronize()
        mo = int(count.item())
        tables.append((nbr.cpu(), parent.cpu(), off.cpu(), out_idx[:mo].cpu(), mo))
    for a, b_ in zip(tables[0][:4], tables[1][:4]):
        assert torch.equal(a, b_)
    assert tables[0][4] == tables[1][4] > 0
    pairs, pn = oracle.indice_pairs_subm(np.ascontiguousarray(idx), batch, shape, 3)
    hp, hn = ops.rulebook_pairs(tables[1][0].to(d), n, flip=True)
    assert np.array_equal(hn.cpu().numpy(), pn) and np.array_equal(hp.cpu().numpy(), pairs)


def test_rulebook_rows_outside_the_declared_grid_get_the_same_table_from_both_builders(native_lib, oracle):
    """Rows whose batch index or coordinates lie outside (batch, spatial_shape) are invalid input for the reference's
    hash too; what this library promises is that the table does not depend on which builder the workspace size selected:
    such a row is never entered into the cell map, its SubM row holds the centre only, its k2 s2 parent is -1, and the
    rows that ARE inside the grid get exactly the table of the input with the outside rows removed."""
    import ctypes as C
    from doda_amd._lib import lib, check
    from tests.util import random_voxels
    d = dev()
    shape, batch = [33, 21, 47], 3
    good = random_voxels(9, 4000, batch, shape)
    bad = np.array([[batch, 1, 1, 1], [-1, 2, 3, 4], [0, -1, 5, 5], [1, 33, 0, 0], [2, 5, 21, 7], [0, 4, 4, 47],
                    [1, 2, -3, 9], [2, 8, 8, -1], [7, 40, 40, 90]], dtype=np.int32)
    rng = np.random.default_rng(4)
    where = np.sort(rng.choice(good.shape[0] + bad.shape[0], bad.shape[0], replace=False))
    idx = np.empty((good.shape[0] + bad.shape[0], 4), dtype=np.int32)
    is_bad = np.zeros(idx.shape[0], dtype=bool)
    is_bad[where] = True
    idx[is_bad], idx[~is_bad] = bad, good
    old_of_new = np.nonzero(~is_bad)[0]
    new_of_old = np.full(idx.shape[0], -1, dtype=np.int64)
    new_of_old[old_of_new] = np.arange(old_of_new.size)
    shape_c = (C.c_int32 * 3)(*shape)
    stream = torch.cuda.current_stream().cuda_stream

    def build(rows, with_grid):
        n = rows.shape[0]
        ind = torch.from_numpy(np.ascontiguousarray(rows)).to(d)
        base = lib().doda_rulebook_workspace_bytes(n)
        out = []
        for coarse in (False, True):
            shp = [(v - 2) // 2 + 1 for v in shape] if coarse else shape
            nbytes = (base + 255) // 256 * 256 + 4 * batch * shp[0] * shp[1] * shp[2] if with_grid else base
            ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
            if not coarse:
                nbr = torch.empty((27, n), dtype=torch.int32, device=d)
                check(lib().doda_rulebook_subm(ind.data_ptr(), n, shape_c, batch, 3, nbr.data_ptr(), n, ws.data_ptr(),
                                               ws.numel(), stream), "doda_rulebook_subm")
                out.append(nbr.cpu().numpy())
            else:
                parent = torch.empty(n, dtype=torch.int32, device=d)
                off = torch.empty(n, dtype=torch.int32, device=d)
                out_idx = torch.zeros((n, 4), dtype=torch.int32, device=d)
                count = torch.zeros(1, dtype=torch.int32, device=d)
                check(lib().doda_rulebook_down2_assign(ind.data_ptr(), n, shape_c, batch, parent.data_ptr(), off.data_ptr(),
                                                       out_idx.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                      "doda_rulebook_down2_assign")
                mo = int(count.item())
                out += [parent.cpu().numpy(), out_idx[:mo].cpu().numpy(), mo]
        # non-cubic SubM: hash builder only
        k = (C.c_int32 * 3)(3, 1, 3)
        ws = torch.empty(base, dtype=torch.uint8, device=d)
        nbg = torch.empty((9, n), dtype=torch.int32, device=d)
        check(lib().doda_rulebook_subm_generic(ind.data_ptr(), n, shape_c, batch, k, nbg.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                               stream), "doda_rulebook_subm_generic")
        out.append(nbg.cpu().numpy())
        return out

    h, g = build(idx, False), build(idx, True)
    for a, b_ in zip(h, g):
        assert np.array_equal(a, b_)
    nbr, parent, out_idx, mo, nbg = h
    centre = np.full(27, -1)
    for t in where:
        centre[13] = t
        assert np.array_equal(nbr[:, t], centre) and parent[t] == -1
        assert np.array_equal(nbg[:, t], np.where(np.arange(9) == 4, t, -1))
    assert not np.isin(nbr[:, ~is_bad], where).any() and not np.isin(nbg[:, ~is_bad], where).any()
    ref_nbr, ref_parent, ref_out_idx, ref_mo, ref_nbg = build(good, True)

    def renumber(tbl):
        return np.where(tbl >= 0, new_of_old[np.maximum(tbl, 0)], -1)
    assert np.array_equal(renumber(nbr[:, ~is_bad]), ref_nbr) and np.array_equal(renumber(nbg[:, ~is_bad]), ref_nbg)
    assert mo == ref_mo and np.array_equal(out_idx, ref_out_idx) and np.array_equal(parent[~is_bad], ref_parent)


@pytest.mark.parametrize("cin,cout,option", [(16, 16, 4), (32, 32, 5)])
def test_pipelined_and_dual_block_tile_kernels_equal_the_plain_tile_kernel_bit_for_bit(native_lib, oracle, cin, cout, option):
    """conv_tile16 (16 -> 16, the next tile prefetched into registers: DODA_OPT_TILE_PIPELINE) and the dual-block pass of the
    32-output-channel layers (DODA_OPT_TILE_DUAL) change WHEN operands are fetched and how many accumulators a pass keeps, not what
    is summed in which order: outputs and per-column statistics totals must equal the plain conv_tile's exactly — forward and data
    gradient, with residual + statistics (the step's instantiation), on a rulebook with tiles that lost their lists."""
    from doda_amd import ops
    from doda_amd._lib import lib
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    n = idx.shape[0]
    tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    g = torch.Generator().manual_seed(77 + option)
    x = torch.randn(n, cin, generator=g).bfloat16().to(d)
    dy = torch.randn(n, cout, generator=g).bfloat16().to(d)
    w = (torch.randn(27, cin, cout, generator=g) * 0.1).to(d)
    assert lib().doda_get_option(option) == 1
    try:
        for inp, layout, nc in ((x, 0, cout), (dy, 2, cin)):
            res = torch.randn(n, nc, generator=g).bfloat16().to(d)
            got = {}
            for on in (1, 0):
                assert lib().doda_set_option(option, on) == 0
                y, st = ops.spconv_gather(inp, w, tbl, n, layout, nc, tilebook=tb, residual=res, want_stats=True)
                yp = ops.spconv_gather(inp, w, tbl, n, layout, nc, tilebook=tb, out_f32=True)
                got[on] = (y, st.double().sum(0).cpu(), yp, st.shape[0])
            assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][2], got[0][2]), (layout, "outputs")
            # (the partial rows are per workgroup — 512 against 768 of them — so only their totals are comparable: fp32 partials
            # summed in fp64 over different groupings of the same values)
            assert rel_err(got[1][1], got[0][1]) < 1e-6, (layout, "statistics")
            if option == 4:
                assert got[1][3] == 512 and got[0][3] == 768
    finally:
        lib().doda_set_option(option, 1)


@pytest.mark.parametrize("tiles,ragged", [(769, 0), (770, 37), (1024, 0), (1025, 255), (1249, 1)])
def test_pipelined_tile_kernel_at_the_edges_of_its_schedule(native_lib, oracle, tiles, ragged):
    """conv_tile16's schedule has corners the two big-scene tests do not reach: workgroups with ONE tile next to workgroups
    with two (769 tiles on 512 workgroups: 'no next tile' requests from the first iteration on), exact multiples of the grid,
    a last tile of a single row / of 255 rows.  Truncated copies of the big scene's gather table (entries past the cut = absent)
    are valid gather tables; the plain conv_tile — oracle-checked above — is the reference, bit for bit, forward with residual +
    statistics and plain with fp32 output."""
    from doda_amd import ops
    from doda_amd._lib import lib
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    tbl_full = ops.rulebook_subm(torch.from_numpy(idx).to(d), shape, batch, 3)
    n = (tiles - 1) * 256 + (ragged if ragged else 256)
    assert n <= tbl_full.shape[1]
    tbl = tbl_full[:, :n].clone()
    tbl[tbl >= n] = -1
    tbl = tbl.contiguous()
    tb = ops.tilebook_build(tbl)
    assert tb is not None
    g = torch.Generator().manual_seed(tiles)
    x = torch.randn(n, 16, generator=g).bfloat16().to(d)
    res = torch.randn(n, 16, generator=g).bfloat16().to(d)
    w = (torch.randn(27, 16, 16, generator=g) * 0.1).to(d)
    got = {}
    try:
        for on in (1, 0):
            assert lib().doda_set_option(4, on) == 0
            y, st = ops.spconv_gather(x, w, tbl, n, 0, 16, tilebook=tb, residual=res, want_stats=True)
            yp = ops.spconv_gather(x, w, tbl, n, 0, 16, tilebook=tb, out_f32=True)
            got[on] = (y, st.double().sum(0).cpu(), yp, st.shape[0])
    finally:
        lib().doda_set_option(4, 1)
    assert got[1][3] == 512 and got[0][3] == min(768, (tiles + 7) // 8 * 8)
    assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][2], got[0][2])
    assert rel_err(got[1][1], got[0][1]) < 1e-6
    # and against the definition on a sample of rows (the truncated table is not an oracle rulebook: fp64 gather-sum here)
    rows = torch.cat([torch.arange(0, 300), torch.arange(n - min(n, 300), n), torch.randint(0, n, (400,), generator=g)]).to(d)
    nb = tbl[:, rows].long()                                             # [27, r]
    xr = torch.cat([x.double(), torch.zeros(1, 16, dtype=torch.float64, device=d)])[nb.clamp(min=-1)]   # -1 -> the zero row
    want = torch.einsum("orc,ock->rk", xr, w.bfloat16().double())
    assert rel_err(got[1][2][rows].cpu(), want.cpu()) < 1e-4


@pytest.mark.parametrize("layout,nc", [(0, 16), (1, 16), (0, 32)])
def test_conv_up32_one_gather_per_output_row(native_lib, oracle, layout, nc):
    """conv_up32 (K = 8 tables read from the fine side: inverse convolution forward, strided convolution data gradient; 32
    input channels): against conv_fast (DODA_OPT_CONV_UP off) bit for bit — a row has one source, so both kernels form the
    same single MFMA product per row and add zeros otherwise — plain, fp32 output, and with residual + statistics; against
    the fp64 definition on sampled rows; and, on a synthetic table whose rows have 0..8 sources (the extra passes of the
    kernel), against conv_fast within the fp32 summation order."""
    from doda_amd import ops
    from doda_amd._lib import lib
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    n = idx.shape[0]
    _, child, par_off, _ = ops.rulebook_down2(torch.from_numpy(idx).to(d), shape, batch)
    m_c = child.shape[1]
    assert m_c < n and par_off.shape == (8, n)
    g = torch.Generator().manual_seed(31 + layout + nc)
    xc = torch.randn(m_c, 32, generator=g).bfloat16().to(d)
    res = torch.randn(n, nc, generator=g).bfloat16().to(d)
    w = (torch.randn(8, 32, nc, generator=g) * 0.1) if layout == 0 else (torch.randn(8, nc, 32, generator=g) * 0.1)
    w = w.to(d)
    # a table with several sources per row: every entry kept with probability 1/2 of a random table into the coarse rows
    multi = torch.randint(0, m_c, (8, n), generator=g, dtype=torch.int32)
    multi[torch.rand(8, n, generator=g) < 0.5] = -1
    multi = multi.to(d)
    got = {}
    try:
        for on in (1, 0):
            assert lib().doda_set_option(6, on) == 0
            y = ops.spconv_gather(xc, w, par_off, n, layout, nc)
            y32 = ops.spconv_gather(xc, w, par_off, n, layout, nc, out_f32=True)
            yr, st = ops.spconv_gather(xc, w, par_off, n, layout, nc, residual=res, want_stats=True)
            ym = ops.spconv_gather(xc, w, multi, n, layout, nc, out_f32=True)
            got[on] = (y, y32, yr, st.double().sum(0).cpu(), ym, st.shape[0])
    finally:
        lib().doda_set_option(6, 1)
    assert got[1][5] == (n + 255) // 256
    assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][1], got[0][1]) and torch.equal(got[1][2], got[0][2])
    assert rel_err(got[1][3], got[0][3]) < 1e-6
    assert rel_err(got[1][4].cpu(), got[0][4].cpu()) < 1e-5
    # fp64 definition on sampled rows (single-source table and the multi-source one)
    rows = torch.cat([torch.arange(0, 512), torch.arange(n - 300, n), torch.randint(0, n, (600,), generator=g)]).to(d)
    wd = w.bfloat16().double() if layout == 0 else w.bfloat16().double().transpose(1, 2)      # [8, 32, nc]
    xz = torch.cat([xc.double(), torch.zeros(1, 32, dtype=torch.float64, device=d)])
    for tbl_k, out in ((par_off, got[1][1]), (multi, got[1][4])):
        nb = tbl_k[:, rows].long()
        want = torch.einsum("orc,ock->rk", xz[nb], wd)
        assert rel_err(out[rows].cpu(), want.cpu()) < 1e-4


def test_dual_pass_tile_kernel_with_four_channel_blocks_vs_oracle(native_lib, oracle):
    """32 -> 64 channels over a tilebook (the data gradient of the level-2 decoder's 64 -> 32 layer has this shape): two dual
    passes of conv_tile<1, ., ., 4, true>, statistics of four channel blocks per workgroup.  Against the oracle's forward on
    bf16-representable operands (one bf16 rounding on the output), statistics totals against fp64 column sums of the stored
    tensor, and against the dense-table kernel (DODA_OPT_TILE_DUAL off sends this shape to conv_fast)."""
    from doda_amd import ops
    from doda_amd._lib import lib
    d = dev()
    idx, shape, batch, pairs, pn = _big_scene(oracle)
    n = idx.shape[0]
    tbl, tb = _hip_rulebook(idx, shape, batch, pairs, pn)
    g = torch.Generator().manual_seed(3264)
    x = torch.randn(n, 32, generator=g).bfloat16()
    w = (torch.randn(3, 3, 3, 32, 64, generator=g) * 0.1).bfloat16().float()
    res = torch.randn(n, 64, generator=g).bfloat16()
    ref = oracle.indice_conv(x.double(), w.double(), pairs, pn, n, False, True) + res.double()
    xd, wd, rd = x.to(d), w.to(d).view(27, 32, 64), res.to(d)
    got = {}
    try:
        for on in (1, 0):
            assert lib().doda_set_option(5, on) == 0
            y, st = ops.spconv_gather(xd, wd, tbl, n, 0, 64, tilebook=tb, residual=rd, want_stats=True)
            got[on] = (y, st.double().sum(0).cpu(), st.shape[0])
    finally:
        lib().doda_set_option(5, 1)
    assert got[1][2] == 512 and got[0][2] > 768          # the tile kernel's one row per workgroup / conv_fast's row per 256 rows
    tol = 2.0 ** -7 * ref.abs() + 1e-5 * float(ref.abs().max())
    for on in (1, 0):
        y = got[on][0]
        assert bool(((y.float().cpu().double() - ref).abs() <= tol).all()), on
        yf = y.double().cpu()
        assert rel_err(got[on][1][0], yf.sum(0)) < 1e-5 and rel_err(got[on][1][1], (yf * yf).sum(0)) < 1e-5, on


def test_restricted_backward_leaves_parameter_gradients_alone(native_lib):
    """with direct parameter gradients on (set_deferred_wgrad(True)) the extension's nodes keep no autograd edge to
    conv weights / BatchNorm vectors and deposit .grad themselves — but only in a plain accumulating backward pass.
    torch.autograd.grad(loss, x) and loss.backward(inputs=[x]) must neither touch .grad nor queue weight-gradient jobs; a
    plain loss.backward() afterwards still produces every gradient (incl. with the coarse levels as one extension call)."""
    ext = _ext_or_skip()
    from doda_amd import model as M
    from doda_amd import spconv
    from doda_amd.model import SparseConvNet, default_cfg
    from doda_amd.spconv import functional as Fsp
    from tests.util import deterministic_init, surface_voxels
    d = dev()
    net = deterministic_init(SparseConvNet(default_cfg()), seed=2).to(d).train()
    ub = net.unet.u.u.u.u                                   # level 5 subtree (ResidualBlocks, strided / inverse convs)
    idx = torch.from_numpy(np.ascontiguousarray(surface_voxels(3, 1500, 2, [32, 32, 32]))).to(d)
    old = (M.COARSE_MODE, M.COARSE_EXEC_LEVEL)
    try:
        assert Fsp.set_deferred_wgrad(True)
        for coarse_mode in ("off", "layers"):
            M.set_coarse_mode(coarse_mode, 5)
            for p in ub.parameters():
                p.grad = None

            def forward():
                x = torch.randn(idx.shape[0], 80, device=d).to(torch.bfloat16).requires_grad_(True)
                t = spconv.SparseConvTensor(x, idx, [32, 32, 32], 2)
                spconv.ops.build_pyramid(t, 3, first_level=5)
                return x, ub(t).features.float().sum()

            x, loss = forward()
            gx, = torch.autograd.grad(loss, x)
            assert gx.shape == x.shape and float(gx.float().abs().sum()) > 0
            assert all(p.grad is None for p in ub.parameters()) and ext.pending_wgrads() == 0
            x, loss = forward()
            loss.backward(inputs=[x])
            assert x.grad is not None and all(p.grad is None for p in ub.parameters()) and ext.pending_wgrads() == 0
            x, loss = forward()
            loss.backward()
            torch.cuda.synchronize()
            assert all(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in ub.parameters())
    finally:
        Fsp.set_deferred_wgrad(False)
        M.set_coarse_mode(*old)
