"""GPU parity tests proper: the HIP path, called through the C ABI (doda_amd.ops -> ctypes ->
libdoda_hip.so), against the CPU oracle on the same seeded inputs.  Integer / index results are
compared bit-exactly; fp32 features to 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from tests.util import random_voxels, surface_voxels

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star: within 1e-4 rel on fp32 features


def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------ voxelisation
def _points(seed, n, batch, extent):
    rng = np.random.default_rng(seed)
    c = rng.integers(0, extent, size=(n, 3))
    b = np.sort(rng.integers(0, batch, size=(n, 1)), axis=0)
    return np.concatenate([b, c], 1).astype(np.int64)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("ncol", [3, 4])
def test_voxelize_idx_host_and_device(native_lib, oracle, mode, ncol):
    from doda_amd import ops
    coords = _points(3 + mode, 5000, 3, 14)
    if ncol == 3:
        coords = coords[:, 1:].copy()
    ref = oracle.voxelize_idx(coords, mode)
    host = ops.voxelize_idx_host(torch.from_numpy(coords), 3, mode)
    devr = ops.voxelize_idx_device(torch.from_numpy(coords).to(dev()), 3, mode)
    for r, h, d in zip(ref, host, devr):
        assert np.array_equal(r, h.numpy())
        assert np.array_equal(r, d.cpu().numpy())


def test_voxelize_idx_mode0_and_empty(native_lib, oracle):
    from doda_amd import ops
    coords = np.unique(_points(11, 800, 2, 30), axis=0)
    np.random.default_rng(0).shuffle(coords)
    ref = oracle.voxelize_idx(coords, 0)
    got = ops.voxelize_idx_device(torch.from_numpy(coords).to(dev()), 2, 0)
    for r, d in zip(ref, got):
        assert np.array_equal(r, d.cpu().numpy())
    empty = torch.zeros((0, 4), dtype=torch.int64)
    oc, im, om = ops.voxelize_idx_host(empty, 1, 4)
    assert oc.shape[0] == 0 and im.shape[0] == 0 and om.shape[0] == 0
    oc, im, om = ops.voxelize_idx_device(empty.to(dev()), 1, 4)
    assert oc.shape[0] == 0 and im.shape[0] == 0 and om.shape[0] == 0


def test_voxelize_idx_device_caller_sizes(native_lib, oracle):
    """sizes=(n_voxels, max_active): exact sizes give the reference result without the read-back; sizes that disagree with
    what stage 1 counted stay inside the outputs:
    too large leaves trailing rows empty, too small keeps a prefix of the voxels and at most max_active points of each."""
    from doda_amd import ops
    coords = _points(5, 6000, 3, 12)
    ref_c, ref_im, ref_om = oracle.voxelize_idx(coords, 4)
    m, width = ref_om.shape
    cd = torch.from_numpy(coords).to(dev())
    oc, im, om = ops.voxelize_idx_device(cd, 3, 4, sizes=(m, width - 1))
    assert np.array_equal(oc.cpu().numpy(), ref_c) and np.array_equal(om.cpu().numpy(), ref_om)
    oc, im, om = ops.voxelize_idx_device(cd, 3, 4, sizes=(m + 37, width + 2))
    om, oc = om.cpu().numpy(), oc.cpu().numpy()
    assert np.array_equal(om[:m, :width], ref_om) and (om[:m, width:] == -1).all()
    assert (om[m:, 0] == 0).all() and (om[m:, 1:] == -1).all() and (oc[m:] == 0).all() and np.array_equal(oc[:m], ref_c)
    assert np.array_equal(im.cpu().numpy(), ref_im)
    small_m, small_a = m - 100, max(width - 3, 1)
    torch.cuda.synchronize()
    oc, im, om = ops.voxelize_idx_device(cd, 3, 4, sizes=(small_m, small_a))
    torch.cuda.synchronize()
    om, oc = om.cpu().numpy(), oc.cpu().numpy()
    assert om.shape == (small_m, small_a + 1) and np.array_equal(om[:, 0], np.minimum(ref_om[:small_m, 0], small_a))
    full = ref_om[:small_m, 0] <= small_a                              # rows that fit are the reference's
    assert np.array_equal(om[full], ref_om[:small_m][full][:, :small_a + 1]) and np.array_equal(oc[full], ref_c[:small_m][full])
    for v in np.nonzero(~full)[0][:50]:                                # the others hold small_a distinct points of the voxel
        got = om[v, 1:]
        assert len(set(got.tolist())) == small_a and set(got.tolist()) <= set(ref_om[v, 1:1 + ref_om[v, 0]].tolist())


@pytest.mark.parametrize("c", [1, 3, 6, 32])
@pytest.mark.parametrize("mode", [3, 4])
def test_voxelize_fp_bp_bit_exact(native_lib, oracle, c, mode):
    from doda_amd import pointgroup_ops as pg
    coords = _points(21, 20000, 4, 20)
    _, _, om = oracle.voxelize_idx(coords, mode)
    rng = np.random.default_rng(5)
    feats = rng.standard_normal((coords.shape[0], c)).astype(np.float32)
    ref = oracle.voxelize_fp(feats, om, average=(mode == 4))
    f = torch.from_numpy(feats).to(dev()).requires_grad_(True)
    rules = torch.from_numpy(om).to(dev())
    out = pg.voxelization(f, rules, mode)
    assert np.array_equal(ref.view(np.uint32), out.detach().cpu().numpy().view(np.uint32))
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev()))
    ref_g = oracle.voxelize_bp(g, om, coords.shape[0], average=(mode == 4))
    assert np.array_equal(ref_g.view(np.uint32), f.grad.cpu().numpy().view(np.uint32))
    # point_recover: voxel -> point broadcast and its backward
    vf = torch.from_numpy(g).to(dev()).requires_grad_(True)
    rec = pg.point_recover(vf, rules, coords.shape[0])
    assert np.array_equal(oracle.voxelize_bp(g, om, coords.shape[0], average=False).view(np.uint32),
                          rec.detach().cpu().numpy().view(np.uint32))


# ------------------------------------------------------------------ rulebooks
CASES = [(0, 700, 2, [16, 12, 20], random_voxels), (1, 3000, 3, [33, 31, 17], surface_voxels),
         (2, 20000, 4, [128, 128, 128], surface_voxels), (3, 1, 1, [4, 4, 4], random_voxels),
         (4, 64, 1, [4, 4, 4], random_voxels)]


@pytest.mark.parametrize("seed,n,batch,shape,gen", CASES)
def test_rulebook_subm_bit_exact(native_lib, oracle, seed, n, batch, shape, gen):
    from doda_amd import spconv
    idx = gen(seed, n, batch, shape)
    ref_pairs, ref_num = oracle.indice_pairs_subm(idx, batch, shape, 3)
    data = spconv.ops.build_subm(torch.from_numpy(idx).to(dev()), batch, shape, 3)
    outids, _, pairs, pair_num, _ = data
    assert np.array_equal(outids.cpu().numpy(), idx)
    assert np.array_equal(pair_num.cpu().numpy(), ref_num)
    assert np.array_equal(pairs.cpu().numpy(), ref_pairs)


@pytest.mark.parametrize("seed,n,batch,shape,gen", CASES)
def test_rulebook_down2_bit_exact(native_lib, oracle, seed, n, batch, shape, gen):
    from doda_amd import spconv
    idx = gen(seed, n, batch, shape)
    ref_out, ref_pairs, ref_num, ref_shape = oracle.indice_pairs_conv(idx, batch, shape, 2, 2, 0, 1)
    data = spconv.ops.build_down2(torch.from_numpy(idx).to(dev()), batch, shape, 2, 2, 0, 1)
    outids, _, pairs, pair_num, _ = data
    assert data.out_spatial_shape == ref_shape
    assert np.array_equal(outids.cpu().numpy(), ref_out)
    assert np.array_equal(pair_num.cpu().numpy(), ref_num)
    assert np.array_equal(pairs.cpu().numpy(), ref_pairs)


def test_rulebook_odd_shape_drops_border(native_lib, oracle):
    """Odd spatial extents: inputs whose cell falls outside (s-2)//2+1 get no pair."""
    from doda_amd import spconv
    shape = [13, 11, 9]
    idx = random_voxels(9, 600, 2, shape)
    ref_out, ref_pairs, ref_num, _ = oracle.indice_pairs_conv(idx, 2, shape, 2, 2, 0, 1)
    assert ref_num.sum() < idx.shape[0]
    data = spconv.ops.build_down2(torch.from_numpy(idx).to(dev()), 2, shape, 2, 2, 0, 1)
    assert np.array_equal(data.indice_pairs.cpu().numpy(), ref_pairs)
    assert np.array_equal(data.outids.cpu().numpy(), ref_out)


# ------------------------------------------------------------------ convolutions
CHANNELS = [(3, 16), (16, 16), (32, 16), (16, 32), (48, 48), (64, 32), (112, 112), (5, 7), (20, 40)]


def _conv_case(oracle, seed, cin, cout, n=1500, batch=2, shape=(24, 20, 22)):
    shape = list(shape)
    idx = surface_voxels(seed, n, batch, shape)
    rng = np.random.default_rng(seed + 100)
    x = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    return idx, shape, batch, x, rng


@pytest.mark.parametrize("cin,cout", CHANNELS)
def test_subm_conv_fwd_bwd(native_lib, oracle, cin, cout):
    from doda_amd import spconv
    idx, shape, batch, x, rng = _conv_case(oracle, cin * 7 + cout, cin, cout)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) * 0.2).astype(np.float32)
    gy = rng.standard_normal((idx.shape[0], cout)).astype(np.float32)
    pairs, pn = oracle.indice_pairs_subm(idx, batch, shape, 3)
    x64, w64, g64 = (torch.from_numpy(a).double() for a in (x, w, gy))
    ref_y = oracle.indice_conv(x64, w64, pairs, pn, idx.shape[0], False, True)
    ref_dx, ref_dw = oracle.indice_conv_backward(x64, w64, g64, pairs, pn, False, True)

    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(dev())
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
    st = spconv.SparseConvTensor(xt, torch.from_numpy(idx).to(dev()), shape, batch)
    out = conv(st)
    out.features.backward(torch.from_numpy(gy).to(dev()))
    assert rel_err(out.features.detach().cpu(), ref_y) < RTOL
    assert rel_err(xt.grad.cpu(), ref_dx) < RTOL
    assert rel_err(conv.weight.grad.cpu(), ref_dw) < RTOL


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 48), (96, 112), (6, 10)])
def test_down2_and_inverse_conv_fwd_bwd(native_lib, oracle, cin, cout):
    from doda_amd import spconv
    idx, shape, batch, x, rng = _conv_case(oracle, cin + cout, cin, cout, shape=(25, 20, 23))
    w = (rng.standard_normal((2, 2, 2, cin, cout)) * 0.3).astype(np.float32)
    wi = (rng.standard_normal((2, 2, 2, cout, cin)) * 0.3).astype(np.float32)
    oi, pairs, pn, oshape = oracle.indice_pairs_conv(idx, batch, shape, 2, 2, 0, 1)
    gy = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    x64, w64, wi64, g64 = (torch.from_numpy(a).double() for a in (x, w, wi, gy))
    ref_mid = oracle.indice_conv(x64, w64, pairs, pn, oi.shape[0], False, False)
    ref_y = oracle.indice_conv(ref_mid, wi64, pairs, pn, idx.shape[0], True, False)
    ref_dmid, ref_dwi = oracle.indice_conv_backward(ref_mid, wi64, g64, pairs, pn, True, False)
    ref_dx, ref_dw = oracle.indice_conv_backward(x64, w64, ref_dmid, pairs, pn, False, False)

    down = spconv.SparseConv3d(cin, cout, kernel_size=2, stride=2, bias=False, indice_key="d").to(dev())
    up = spconv.SparseInverseConv3d(cout, cin, kernel_size=2, bias=False, indice_key="d").to(dev())
    with torch.no_grad():
        down.weight.copy_(torch.from_numpy(w))
        up.weight.copy_(torch.from_numpy(wi))
    xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
    st = spconv.SparseConvTensor(xt, torch.from_numpy(idx).to(dev()), shape, batch)
    mid = down(st)
    assert mid.spatial_shape == oshape
    assert np.array_equal(mid.indices.cpu().numpy(), oi)
    out = up(mid)
    assert np.array_equal(out.indices.cpu().numpy(), idx) and out.spatial_shape == shape
    out.features.backward(torch.from_numpy(gy).to(dev()))
    assert rel_err(mid.features.detach().cpu(), ref_mid) < RTOL
    assert rel_err(out.features.detach().cpu(), ref_y) < RTOL
    assert rel_err(xt.grad.cpu(), ref_dx) < RTOL
    assert rel_err(down.weight.grad.cpu(), ref_dw) < RTOL
    assert rel_err(up.weight.grad.cpu(), ref_dwi) < RTOL


def test_subm_conv_matches_dense_definition(native_lib):
    """The HIP path against the definitional dense conv3d (fp64), independent of the rulebook
    restatement."""
    from doda_amd import spconv
    from oracle import dense_ref
    shape, batch = [14, 12, 10], 2
    idx = random_voxels(4, 500, batch, shape)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((idx.shape[0], 16)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 16, 32)) * 0.2).astype(np.float32)
    ref = dense_ref.subm_conv(torch.from_numpy(x).double(), idx, shape, batch, torch.from_numpy(w).double())
    conv = spconv.SubMConv3d(16, 32, 3, padding=1, bias=False).to(dev())
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    st = spconv.SparseConvTensor(torch.from_numpy(x).to(dev()), torch.from_numpy(idx).to(dev()), shape, batch)
    assert rel_err(conv(st).features.detach().cpu(), ref) < RTOL


def test_conv1x1_and_empty(native_lib):
    from doda_amd import spconv
    conv = spconv.SubMConv3d(32, 16, kernel_size=1, bias=False).to(dev())
    x = torch.randn(100, 32, device=dev())
    idx = torch.from_numpy(random_voxels(1, 100, 1, [8, 8, 8])).to(dev())
    out = conv(spconv.SparseConvTensor(x, idx, [8, 8, 8], 1))
    assert torch.allclose(out.features, x @ conv.weight.view(32, 16), rtol=1e-5, atol=1e-5)
    conv3 = spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="e").to(dev())
    e = spconv.SparseConvTensor(torch.zeros(0, 16, device=dev()),
                                torch.zeros((0, 4), dtype=torch.int32, device=dev()), [8, 8, 8], 1)
    assert conv3(e).features.shape == (0, 16)


def test_maxpool_fwd_bwd(native_lib, oracle):
    from doda_amd import spconv
    shape, batch = [10, 12, 8], 2
    idx = random_voxels(6, 400, batch, shape)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((idx.shape[0], 8)).astype(np.float32)
    oi, pairs, pn, _ = oracle.indice_pairs_conv(idx, batch, shape, 2, 2, 0, 1)
    ref = oracle.indice_maxpool(torch.from_numpy(x), pairs, pn, oi.shape[0])
    pool = spconv.SparseMaxPool3d(2, 2)
    xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
    out = pool(spconv.SparseConvTensor(xt, torch.from_numpy(idx).to(dev()), shape, batch))
    assert np.array_equal(out.features.detach().cpu().numpy(), ref.numpy())
    out.features.sum().backward()
    # each output channel routes its unit gradient to the arg-max input(s)
    g = xt.grad.cpu().numpy()
    assert g.min() >= 0 and abs(g.sum() - (ref.numpy() > 0).sum()) < 1e-3


# ------------------------------------------------------------------ neighbour queries
def _clouds(seed, sizes, lattice=False):
    rng = np.random.default_rng(seed)
    pts = []
    for n in sizes:
        if lattice:
            pts.append(rng.integers(0, 4, size=(n, 3)).astype(np.float32))  # many exact ties
        else:
            pts.append(rng.standard_normal((n, 3)).astype(np.float32))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    return np.concatenate(pts, 0), off


@pytest.mark.parametrize("k", [1, 3, 16, 100])
@pytest.mark.parametrize("lattice", [False, True])
def test_knnquery_bit_exact(native_lib, oracle, k, lattice):
    from doda_amd import pointops2
    xyz, off = _clouds(1, [700, 300, 1200], lattice)
    new_xyz, noff = _clouds(2, [900, 100, 500], lattice)
    ref_idx, ref_d2 = oracle.knnquery(k, xyz, new_xyz, off[1:], noff[1:])
    d = dev()
    idx, dist = pointops2.knnquery(k, torch.from_numpy(xyz).to(d), torch.from_numpy(new_xyz).to(d),
                                   torch.from_numpy(off).to(d), torch.from_numpy(noff).to(d))
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(ref_d2))


def test_knnquery_survey_known_answer(native_lib):
    """recorded from the reference kernel body (heap order on a 3-way tie)."""
    from doda_amd import pointops2
    d = dev()
    xyz = torch.tensor([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [5, 5, 5]], dtype=torch.float32, device=d)
    q = torch.tensor([[0, 0, 0], [0.5, 0, 0]], dtype=torch.float32, device=d)
    idx, dist = pointops2.knnquery(3, xyz, q, torch.tensor([0, 5], dtype=torch.int32, device=d),
                                   torch.tensor([0, 2], dtype=torch.int32, device=d))
    assert idx.cpu().tolist() == [[0, 2, 1], [1, 0, 3]]
    assert torch.allclose(dist.cpu() ** 2, torch.tensor([[0, 1, 1], [0.25, 0.25, 1.25]]))


@pytest.mark.parametrize("k", [1, 8, 40])
def test_knn_batch_bit_exact(native_lib, oracle, k):
    from doda_amd import pointgroup_ops as pg
    xyz, off = _clouds(3, [400, 600], True)
    qxyz, qoff = _clouds(4, [500, 300], True)
    bi = np.repeat(np.arange(2), np.diff(off)).astype(np.int32)
    ref = oracle.knn_batch(xyz, qxyz, bi, qoff, k)
    d = dev()
    got = pg.knn(torch.from_numpy(xyz).to(d), torch.from_numpy(qxyz).to(d), torch.from_numpy(bi).to(d),
                 torch.from_numpy(qoff).to(d), k)
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("mean_active", [2, 50])
def test_ballquery_bit_exact(native_lib, oracle, mean_active):
    from doda_amd import pointgroup_ops as pg
    xyz, off = _clouds(5, [800, 500])
    bi = np.repeat(np.arange(2), np.diff(off)).astype(np.int32)
    d = dev()
    idx, start_len = pg.ballquery_batch_p(torch.from_numpy(xyz).to(d), torch.from_numpy(bi).to(d),
                                          torch.from_numpy(off).to(d), 0.6, mean_active)
    # replay the wrapper's grow-and-retry loop on the oracle
    ma = mean_active
    while True:
        ref_idx, ref_sl, total = oracle.ballquery(xyz, bi, off, 0.6, ma)
        if total <= xyz.shape[0] * ma:
            break
        ma = int(total // xyz.shape[0] + 1)
    assert np.array_equal(start_len.cpu().numpy(), ref_sl)
    assert np.array_equal(idx.cpu().numpy(), ref_idx[:total])


# ------------------------------------------------------------------ fused BatchNorm(+ReLU)
@pytest.mark.parametrize("c", [16, 48, 112, 192])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_relu_matches_torch(native_lib, c, relu, dtype):
    """doda_amd.nn against torch.nn.BatchNorm1d(+ReLU) evaluated in fp64 (training and eval)."""
    from doda_amd import nn as dnn
    torch.manual_seed(c)
    d = dev()
    for m in (5000 + c, 40000 + c):   # single-launch small-M kernels and the multi-block path
        _check_bn(dnn, m, c, relu, dtype, d)


def _check_bn(dnn, m, c, relu, dtype, d):
    x = (torch.randn(m, c, device=d) * 1.7 + 3.0).to(dtype)       # non-zero mean: exercises the shift
    gy = torch.randn(m, c, device=d).to(dtype)
    ref = torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1).to(d).double()
    mine = torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1).to(d)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(c) + 0.5); ref.bias.copy_(torch.randn(c) * 0.3)
        mine.weight.copy_(ref.weight.float()); mine.bias.copy_(ref.bias.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    for training in (True, False):
        ref.train(training); mine.train(training)
        xm = x.clone().requires_grad_(True)
        assert dnn.fusable(mine, xm)
        ym = dnn.batch_norm_relu(xm, mine, relu)
        ym.backward(gy)
        xr = x.double().requires_grad_(True)
        yr = ref(xr)
        assert rel_err(ym.detach().float().cpu(), (torch.relu(yr) if relu else yr).detach().cpu()) < tol
        if relu:
            # backward reference uses the kernel's own activation pattern: the mask is recomputed in
            # fp32, so a pre-activation within rounding of 0 may legitimately fall on the other side
            # than in fp64 (a few of 4.5 M elements), which would otherwise dominate max-norm errors
            yr = yr * (ym.detach() > 0).double()
        yr.backward(gy.double())
        assert rel_err(xm.grad.float().cpu(), xr.grad.cpu()) < tol
        assert rel_err(mine.weight.grad.cpu(), ref.weight.grad.cpu()) < tol
        assert rel_err(mine.bias.grad.cpu(), ref.bias.grad.cpu()) < tol
        mine.weight.grad = None; mine.bias.grad = None; ref.weight.grad = None; ref.bias.grad = None
    assert rel_err(mine.running_mean.cpu(), ref.running_mean.cpu()) < 1e-4
    assert rel_err(mine.running_var.cpu(), ref.running_var.cpu()) < 1e-3 if dtype == torch.bfloat16 else 1e-4
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_kernels_bitwise_repeatable(native_lib, dtype):
    """Race screen for the hand-pipelined kernels (inline-asm loads, counted vmcnt, wave-private LDS):
    the same launch repeated under changing machine load must return bit-identical results, across
    tile shapes (row counts select different S/NBW/D instantiations) and all three weight layouts."""
    from doda_amd import ops, spconv
    d = dev()
    for m, c_in, c_out, seed in ((70000, 16, 16, 0), (9000, 32, 48, 1), (700, 96, 112, 2), (150, 48, 16, 3)):
        shape = [64, 64, 48]
        idx = surface_voxels(seed, m, 2, shape)
        n = idx.shape[0]
        sub = spconv.ops.build_subm(torch.from_numpy(idx).to(d), 2, shape, 3)
        dn = spconv.ops.build_down2(torch.from_numpy(idx).to(d), 2, shape, 2, 2, 0, 1)
        x = torch.randn(n, c_in, device=d).to(dtype)
        gy = torch.randn(n, c_out, device=d).to(dtype)
        w = torch.randn(27, c_in, c_out, device=d) * 0.1
        w8 = torch.randn(8, c_in, c_out, device=d) * 0.1
        gmid = torch.randn(dn.outids.shape[0], c_out, device=d).to(dtype)
        fns = [lambda: ops.spconv_gather(x, w, sub.tbl, n, 0, c_out),
               lambda: ops.spconv_gather(gy, w, sub.tbl, n, 2, c_in),
               lambda: ops.spconv_wgrad(x, gy, sub.tbl, n),
               lambda: ops.spconv_gather(x, w8, dn.tbl, dn.outids.shape[0], 0, c_out),
               lambda: ops.spconv_gather(gmid, w8, dn.tbl_rev, n, 1, c_in),
               lambda: ops.spconv_wgrad(x, gmid, dn.tbl, dn.outids.shape[0])]
        for fn in fns:
            first = fn().clone()
            for rep in range(12):
                junk = torch.randn(1 << (14 + rep % 6), device=d).sum()  # perturb timing / occupancy
                assert torch.equal(fn(), first)
            del junk


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m,c_in,c_out", [(60000, 16, 16), (5000, 32, 48), (300, 96, 112), (2000, 6, 10)])
def test_conv_residual_epilogue(native_lib, dtype, m, c_in, c_out):
    """doda_spconv_gather_add_*: y = conv + res with the add fused into the store, against the
    unfused sequence, through the raw op and through the module + autograd (the residual's gradient
    is the incoming gradient; feature / weight gradients are unchanged)."""
    from doda_amd import ops, spconv
    d = dev()
    shape = [64, 64, 48]
    idx = surface_voxels(5, m, 2, shape)
    n = idx.shape[0]
    ind = torch.from_numpy(idx).to(d)
    sub = spconv.ops.build_subm(ind, 2, shape, 3)
    x = torch.randn(n, c_in, device=d).to(dtype)
    res = torch.randn(n, c_out, device=d).to(dtype)
    w = torch.randn(27, c_in, c_out, device=d) * 0.1
    plain = ops.spconv_gather(x, w, sub.tbl, n, 0, c_out)
    fused = ops.spconv_gather(x, w, sub.tbl, n, 0, c_out, residual=res)
    if dtype == torch.float32:
        assert torch.equal(fused, plain + res)          # same fp32 add, same order
    else:                                               # one rounding instead of two
        ref = plain.float() + res.float()
        assert float((fused.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())

    conv = spconv.SubMConv3d(c_in, c_out, 3, bias=False, indice_key="s").to(d)
    grads = []
    for use_res in (False, True):
        xin = x.clone().requires_grad_(True)
        rin = res.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        t = spconv.SparseConvTensor(xin, ind, shape, 2)
        y = conv(t, residual=rin).features if use_res else conv(t).features + rin
        (y.float() * torch.linspace(-1, 1, c_out, device=d)).sum().backward()
        grads.append((y.detach().float(), xin.grad.float(), rin.grad.float(), conv.weight.grad.float().clone()))
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= tol * max(float(a.abs().max()), 1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_multi_matches_per_layer(native_lib, dtype):
    """doda_spconv_wgrad_multi: many layers (mixed sizes, channel counts, SubM and k2s2 tables, several
    kernel variants, single- and multi-chunk jobs) in one call against the per-layer entry point."""
    from doda_amd import ops, spconv
    d = dev()
    jobs, refs = [], []
    for m, c_in, c_out, seed in ((50000, 16, 16, 0), (9000, 32, 32, 1), (9000, 32, 48, 2), (9000, 48, 48, 7), (700, 96, 112, 3),
                                 (150, 48, 16, 4), (50000, 16, 32, 5), (40, 64, 64, 6)):
        shape = [64, 64, 48]
        idx = surface_voxels(seed, m, 2, shape)
        n = idx.shape[0]
        ind = torch.from_numpy(idx).to(d)
        sub = spconv.ops.build_subm(ind, 2, shape, 3)
        dn = spconv.ops.build_down2(ind, 2, shape, 2, 2, 0, 1)
        x = torch.randn(n, c_in, device=d).to(dtype)
        gy = torch.randn(n, c_out, device=d).to(dtype)
        gmid = torch.randn(dn.outids.shape[0], c_out, device=d).to(dtype)
        jobs += [(x, gy, sub.tbl, n), (x, gmid, dn.tbl, dn.outids.shape[0])]
        refs += [ops.spconv_wgrad(x, gy, sub.tbl, n), ops.spconv_wgrad(x, gmid, dn.tbl, dn.outids.shape[0])]
    outs = ops.spconv_wgrad_multi(jobs)
    assert len(outs) == len(refs)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape
        assert rel_err(o.cpu(), r.cpu()) < 2e-6        # same products, different partial-sum order
    again = ops.spconv_wgrad_multi(jobs)
    for o, r in zip(outs, again):
        assert torch.equal(o, r)                       # deterministic


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dsnorm_fused_path(native_lib, dtype):
    """DSNorm (reference model/dsnorm.py): the fused BN+ReLU path picks the running statistics of the
    current domain and leaves the other domain's untouched; values and gradients as torch's."""
    from doda_amd import spconv
    from doda_amd.dsnorm import DSNorm1d
    from doda_amd.nn import fusable
    d = dev()
    c, m = 32, 5000
    torch.manual_seed(3)
    mine, ref = DSNorm1d(c, eps=1e-4, momentum=0.1).to(d), DSNorm1d(c, eps=1e-4, momentum=0.1).to(d)
    with torch.no_grad():
        mine.weight.uniform_(0.5, 1.5); mine.bias.uniform_(-0.5, 0.5)
        ref.load_state_dict(mine.state_dict())
    seq = spconv.SparseSequential(mine, torch.nn.ReLU())
    idx = torch.from_numpy(surface_voxels(1, m, 1, [64, 64, 48])).to(d)
    n = idx.shape[0]
    for label in (1, 0, 1):
        mine.set_domain_label(label); ref.set_domain_label(label)
        x = (torch.randn(n, c, device=d) * 2 + 1).to(dtype)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        assert fusable(mine, xa)
        ya = seq(spconv.SparseConvTensor(xa, idx, [64, 64, 48], 1)).features
        yb = torch.relu(ref(xb.float()))
        g = torch.randn_like(yb)
        (ya.float() * g).sum().backward(); (yb * g).sum().backward()
        tol = 1e-4 if dtype == torch.float32 else 3e-2
        assert rel_err(ya.detach().float().cpu(), yb.detach().cpu()) < tol
        assert rel_err(xa.grad.float().cpu(), xb.grad.float().cpu()) < (1e-3 if dtype == torch.float32 else 5e-2)
    for name in ("running_mean_source", "running_var_source", "running_mean_target", "running_var_target"):
        assert rel_err(getattr(mine, name).cpu(), getattr(ref, name).cpu()) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 3


# ------------------------------------------------------------------ generic geometry
GEOMS = [((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
         ((3, 3, 3), (1, 1, 1), (0, 0, 0), (1, 1, 1)), ((2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1)),
         ((3, 1, 2), (1, 1, 2), (1, 0, 0), (1, 1, 1)), ((3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2)),
         ((2, 3, 1), (2, 1, 1), (0, 1, 0), (1, 1, 1))]


@pytest.mark.parametrize("k,s,p,d", GEOMS)
@pytest.mark.parametrize("seed,n,batch,shape,gen", CASES[:3])
def test_rulebook_generic_conv_bit_exact(native_lib, oracle, seed, n, batch, shape, gen, k, s, p, d):
    """doda_rulebook_conv_*: output numbering, pair lists and pair counts equal to the serial spconv
    algorithm (oracle getIndicePairsConv) for strided, padded, dilated and non-cubic kernels."""
    from doda_amd import spconv
    idx = gen(seed, n, batch, shape)
    ref_out, ref_pairs, ref_num, ref_shape = oracle.indice_pairs_conv(idx, batch, shape, list(k), list(s), list(p), list(d))
    outids, pairs, pair_num = spconv.ops.get_indice_pairs(torch.from_numpy(idx).to(dev()), batch, shape, list(k),
                                                          list(s), list(p), list(d))
    assert np.array_equal(outids.cpu().numpy(), ref_out)
    assert np.array_equal(pair_num.cpu().numpy(), ref_num)
    assert np.array_equal(pairs.cpu().numpy()[:, :, :idx.shape[0]], ref_pairs)


@pytest.mark.parametrize("k", [(3, 1, 3), (1, 3, 1), (3, 3, 1)])
def test_rulebook_generic_subm_bit_exact(native_lib, oracle, k):
    from doda_amd import spconv
    shape, batch = [33, 31, 17], 3
    idx = surface_voxels(1, 3000, batch, shape)
    ref_pairs, ref_num = oracle.indice_pairs_subm(idx, batch, shape, list(k))
    data = spconv.ops.build_subm(torch.from_numpy(idx).to(dev()), batch, shape, list(k))
    assert np.array_equal(data.indice_pair_num.cpu().numpy(), ref_num)
    assert np.array_equal(data.indice_pairs.cpu().numpy(), ref_pairs)


@pytest.mark.parametrize("k,s,p,d", [GEOMS[0], GEOMS[1], GEOMS[4], GEOMS[5]])
def test_generic_conv_fwd_bwd_and_inverse(native_lib, oracle, k, s, p, d):
    """SparseConv3d / SparseInverseConv3d with a generic geometry through the same native conv kernels."""
    from doda_amd import spconv
    cin, cout = 16, 32
    idx, shape, batch, x, rng = _conv_case(oracle, 77, cin, cout, n=1200, shape=(21, 18, 20))
    w = (rng.standard_normal((*k, cin, cout)) * 0.3).astype(np.float32)
    wi = (rng.standard_normal((*k, cout, cin)) * 0.3).astype(np.float32)
    oi, pairs, pn, oshape = oracle.indice_pairs_conv(idx, batch, shape, list(k), list(s), list(p), list(d))
    gy = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    x64, w64, wi64, g64 = (torch.from_numpy(a).double() for a in (x, w, wi, gy))
    ref_mid = oracle.indice_conv(x64, w64, pairs, pn, oi.shape[0], False, False)
    ref_y = oracle.indice_conv(ref_mid, wi64, pairs, pn, idx.shape[0], True, False)
    ref_dmid, ref_dwi = oracle.indice_conv_backward(ref_mid, wi64, g64, pairs, pn, True, False)
    ref_dx, ref_dw = oracle.indice_conv_backward(x64, w64, ref_dmid, pairs, pn, False, False)
    down = spconv.SparseConv3d(cin, cout, kernel_size=list(k), stride=list(s), padding=list(p), dilation=list(d),
                               bias=False, indice_key="g").to(dev())
    up = spconv.SparseInverseConv3d(cout, cin, kernel_size=list(k), bias=False, indice_key="g").to(dev())
    with torch.no_grad():
        down.weight.copy_(torch.from_numpy(w))
        up.weight.copy_(torch.from_numpy(wi))
    xt = torch.from_numpy(x).to(dev()).requires_grad_(True)
    mid = down(spconv.SparseConvTensor(xt, torch.from_numpy(idx).to(dev()), shape, batch))
    assert mid.spatial_shape == oshape and np.array_equal(mid.indices.cpu().numpy(), oi)
    out = up(mid)
    assert np.array_equal(out.indices.cpu().numpy(), idx)
    out.features.backward(torch.from_numpy(gy).to(dev()))
    assert rel_err(mid.features.detach().cpu(), ref_mid) < RTOL
    assert rel_err(out.features.detach().cpu(), ref_y) < RTOL
    assert rel_err(xt.grad.cpu(), ref_dx) < RTOL
    assert rel_err(down.weight.grad.cpu(), ref_dw) < RTOL
    assert rel_err(up.weight.grad.cpu(), ref_dwi) < RTOL


@pytest.mark.parametrize("c", [20, 13, 64])
def test_fused_cross_entropy(native_lib, c):
    """doda_cross_entropy_fwd/_bwd == nn.CrossEntropyLoss(ignore_index=255): value, gradient, all-ignored case."""
    from doda_amd.model import cross_entropy
    d = dev()
    torch.manual_seed(c)
    n = 70001
    logits = (torch.randn(n, c, device=d) * 3).requires_grad_(True)
    labels = torch.randint(0, c, (n,), device=d)
    labels[torch.rand(n, device=d) < 0.2] = 255
    ref_in = logits.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, labels, ignore_index=255)
    got = cross_entropy(logits, labels, ignore_index=255)
    (got * 1.7).backward(); (ref * 1.7).backward()
    assert abs(float(got) - float(ref)) < 2e-6 * max(abs(float(ref)), 1.0)
    assert rel_err(logits.grad.cpu(), ref_in.grad.cpu()) < 2e-6
    again = cross_entropy(logits.detach(), labels, ignore_index=255)
    assert float(again) == float(got)                       # deterministic
    none = cross_entropy(logits.detach(), torch.full_like(labels, 255), ignore_index=255)
    assert float(none) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("m", [300, 30000])
def test_residual_block_identity_skip_fusions(native_lib, dtype, m):
    """SparseSequential(..., residual="input"): skip add in the last conv's store and the skip's gradient in
    the first BatchNorm's backward (doda_bn_relu_bwd_add), against the unfused formulation."""
    from doda_amd import spconv
    d = dev()
    c, shape = 32, [64, 64, 48]
    idx = torch.from_numpy(surface_voxels(2, m, 2, shape)).to(d)
    n = idx.shape[0]
    torch.manual_seed(1)
    seq = spconv.SparseSequential(
        torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1), torch.nn.ReLU(), spconv.SubMConv3d(c, c, 3, bias=False, indice_key="k"),
        torch.nn.BatchNorm1d(c, eps=1e-4, momentum=0.1), torch.nn.ReLU(), spconv.SubMConv3d(c, c, 3, bias=False, indice_key="k")).to(d)
    x0 = (torch.randn(n, c, device=d) * 2).to(dtype)
    g = torch.randn(n, c, device=d)
    res = []
    for fused in (True, False):
        for mod in seq.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.reset_running_stats()
        seq.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        t = spconv.SparseConvTensor(x, idx, shape, 2)
        if fused:
            y = seq(t, residual="input").features
        else:
            y = seq(t).features + x
        (y.float() * g).sum().backward()
        res.append([y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in seq.parameters()])
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)
