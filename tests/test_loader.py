"""doda_amd.loader: the non-blocking loader behind `python -m doda_amd.train` (reference dataset/__init__.py:62-75:
DataLoader + DistributedSampler + collate_fn in worker processes; dataset/dataset.py:121-187: the collate contract)."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def cache(tmp_path_factory):
    from doda_amd.loader import prepare_cache
    d = str(tmp_path_factory.mktemp("scenes"))
    _, paths = prepare_cache(3, 6000, 50, 1000, d, procs=1)
    return paths


def test_epoch_sampler_partitions_like_distributed_sampler():
    from doda_amd.loader import EpochSampler
    world, bs, n = 3, 2, 50
    per_rank = []
    for r in range(world):
        s = EpochSampler(n, bs, r, world, shuffle=True, seed=5)
        s.set_epoch(4)
        idx = list(s)
        assert len(idx) == len(s) and len(idx) % bs == 0 and len(idx) == (n // world if r >= n % world else n // world + 1) // bs * bs
        assert all(4 * n <= i < 5 * n for i in idx)            # global indices of epoch 4
        per_rank.append([i - 4 * n for i in idx])
    flat = sum(per_rank, [])
    assert len(set(flat)) == len(flat)                          # ranks are disjoint
    s2 = EpochSampler(n, bs, 0, world, shuffle=True, seed=5)
    s2.set_epoch(5)
    assert [i - 5 * n for i in s2] != per_rank[0]               # another epoch, another permutation
    s3 = EpochSampler(n, bs, 1, world, shuffle=False)
    assert list(s3) == list(range(1, n, world))[:len(list(s3))]


def test_synthetic_scenes_are_seeded_and_keep_the_dataset_contract(cache):
    from doda_amd.loader import SyntheticScenes, host_collate
    ds = SyntheticScenes(cache, 12, 50, seed=7)
    a, b = ds[5], ds[5]
    assert all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == 5
    c = ds[5 + 12]                                              # same base scene, next epoch: another augmentation
    assert c[0].shape == a[0].shape and not torch.equal(c[1], a[1])
    xyz, mid, lab, _ = a
    assert xyz.dtype == torch.int32 and mid.dtype == torch.float32 and lab.dtype == torch.int32
    assert xyz.min(0)[0].tolist() == [0, 0, 0]                  # xyz * scale - min, truncated (dataset/scannet.py:76-78)
    q = mid * 50.0
    assert torch.equal((q - q.min(0)[0]).to(torch.int32), xyz)
    plain = SyntheticScenes(cache, 12, 50, seed=7, augment=False)[5]
    with np.load(cache[5 % 3]) as f:
        assert torch.equal(plain[1], torch.from_numpy(f["xyz_mid"]))
    # rigid: pairwise distances of the augmented cloud equal the base cloud's up to the +-5 mm jitter
    i, j = 10, 2000
    d0 = (plain[1][i] - plain[1][j]).norm()
    d1 = (mid[i] - mid[j]).norm()
    assert abs(float(d0 - d1)) < 0.02
    hb = host_collate([ds[0], ds[1], ds[2]])
    n = [ds[k][0].shape[0] for k in range(3)]
    assert hb["offsets"].tolist() == [0, n[0], n[0] + n[1], sum(n)]
    assert hb["locs32"].shape == (sum(n), 4) and hb["locs32"][n[0], 0] == 1 and torch.equal(hb["locs32"][:n[0], 1:], ds[0][0])
    top = max(int(ds[k][0].max()) for k in range(3)) + 1
    assert hb["spatial_shape"].max() == max(top, 128) and hb["spatial_shape"].min() >= 128


def test_host_loader_worker_processes_deliver_every_batch_once(cache):
    from doda_amd.loader import SyntheticScenes, host_loader
    ds = SyntheticScenes(cache, 12, 50, seed=3)
    seen = []
    for rank in range(2):
        dl, sampler = host_loader(ds, 2, rank, 2, workers=2, shuffle=True, seed=1)
        sampler.set_epoch(0)
        for hb in dl:
            assert hb["offsets"].numel() == 3 and hb["locs32"].dtype == torch.int32
            seen += hb["id"]
        del dl
    assert sorted(seen) == list(range(12))


@pytest.mark.gpu
def test_device_collate_from_host_concat_equals_per_scene_collate(cache):
    """collate_device_concat(host_collate(items)) == collate_device(items), key by key (the reference's dictionary)."""
    from doda_amd.collate import collate_device, collate_device_concat
    from doda_amd.loader import SyntheticScenes, host_collate
    ds = SyntheticScenes(cache, 12, 50, seed=9)
    items = [ds[k] for k in (3, 4, 8)]
    d = torch.device("cuda:0")
    a = collate_device([(x.numpy().astype(np.int64), m.numpy(), l.numpy().astype(np.int64), i) for x, m, l, i in items], d)
    b = collate_device_concat(host_collate(items), d)
    for k in ("locs", "voxel_locs", "p2v_map", "v2p_map", "v2p_map_t", "locs_float", "feats", "labels"):
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    assert torch.equal(a["offsets"], b["offsets"]) and np.array_equal(a["spatial_shape"], b["spatial_shape"]) and a["id"] == b["id"]


@pytest.mark.gpu
def test_hbm_resident_dataset_feeds_the_trainer_contract(cache):
    """DeviceScenes (dataset in HBM, augmentation on the device) through DeviceFeeder: batches obey the collate contract, the
    sampler's order, the rigid-augmentation property, and arrive with their rulebooks."""
    from doda_amd.loader import DeviceFeeder, DeviceScenes
    from doda_amd.model import PyramidPrefetcher
    d = torch.device("cuda:0")
    dsc = DeviceScenes(cache, 12, 50, seed=11, batch_size=2, rank=0, world=1, device=d)
    dsc.set_epoch(2)
    pf = PyramidPrefetcher(d, 7)
    feeder = DeviceFeeder(dsc, d, prefetcher=pf, with_pairs=False, with_tiles=0)
    ids = []
    try:
        for batch, pyramid in feeder:
            ids += batch["id"]
            n = batch["locs"].shape[0]
            assert batch["locs"].dtype == torch.int64 and batch["offsets"][-1] == n and batch["labels"].shape == (n,)
            assert batch["locs"][:, 1:].min() == 0 and int(batch["locs"][:, 0].max()) == 1
            assert batch["p2v_map"].shape == (n,) and batch["v2p_map"].shape[0] == batch["voxel_locs"].shape[0]
            q = batch["locs_float"] * 50.0
            for b in range(2):
                lo, hi = int(batch["offsets"][b]), int(batch["offsets"][b + 1])
                assert torch.equal((q[lo:hi] - q[lo:hi].min(0)[0]).long(), batch["locs"][lo:hi, 1:])
            idx32, book = pyramid
            assert idx32.shape[0] == batch["voxel_locs"].shape[0] and "subm1" in book and "spconv6" in book
    finally:
        feeder.close()
        pf.shutdown()
    assert sorted(i - 24 for i in ids) == list(range(12))


def test_reorder_voxels_renames_rows_and_keeps_every_point_in_its_voxel():
    """collate.reorder_voxels (Z-order numbering): the two maps that name voxel rows stay consistent — every point still maps to
    the voxel with ITS coordinates, every voxel row lists exactly its points, v2p_map_t stays the transposed table, rows are
    sorted by (scene, Morton key) — and 'first' returns the batch untouched."""
    from doda_amd.collate import morton_keys, reorder_voxels
    from doda_amd.scene import make_batch
    b = make_batch(3, 4000, 11)
    assert reorder_voxels(b, "first") is b
    r = reorder_voxels(b, "morton")
    m = b["voxel_locs"].shape[0]
    assert r["voxel_locs"].shape == b["voxel_locs"].shape and r["p2v_map"].dtype == b["p2v_map"].dtype
    assert torch.equal(torch.sort(morton_keys(b["voxel_locs"]))[0], morton_keys(r["voxel_locs"]))   # a permutation, sorted
    assert torch.equal(b["voxel_locs"][b["p2v_map"].long()], r["voxel_locs"][r["p2v_map"].long()])
    cnt = r["v2p_map"][:, 0].long()
    assert torch.equal(torch.sort(cnt)[0], torch.sort(b["v2p_map"][:, 0].long())[0]) and int(cnt.sum()) == b["p2v_map"].numel()
    rows = torch.repeat_interleave(torch.arange(m), cnt)
    pts = torch.cat([r["v2p_map"][j, 1:1 + int(cnt[j])] for j in range(m)]).long()
    assert torch.equal(r["p2v_map"][pts].long(), rows)
    assert torch.equal(r["v2p_map_t"], r["v2p_map"][:, 1:].t())
    assert (r["voxel_locs"][1:, 0] >= r["voxel_locs"][:-1, 0]).all()        # scenes stay contiguous
    # what the renumbering is for: the 3 x 3 x 3 neighbourhoods of 256 consecutive rows hold fewer DISTINCT voxels
    def distinct_per_tile(v):
        v = v.numpy().astype(np.int64)
        key = lambda c: ((c[:, 0] * 4096 + c[:, 1] + 1) * 4096 + c[:, 2] + 1) * 4096 + c[:, 3] + 1
        present = np.sort(key(v))
        offs = np.array([[0, a, b_, c] for a in (-1, 0, 1) for b_ in (-1, 0, 1) for c in (-1, 0, 1)])
        out = []
        for t0 in range(0, v.shape[0] - 255, 256):
            k = np.unique(key((v[t0:t0 + 256, None, :] + offs[None]).reshape(-1, 4)))
            out.append(int(np.isin(k, present, assume_unique=True).sum()))
        return float(np.median(out))
    big = make_batch(1, 40000, 12)          # (a scene large enough for its scan order to show: strips against patches)
    assert distinct_per_tile(reorder_voxels(big, "morton")["voxel_locs"]) < 0.9 * distinct_per_tile(big["voxel_locs"])
    with pytest.raises(ValueError):
        reorder_voxels(b, "hilbert")


@pytest.mark.gpu
@pytest.mark.parametrize("deferred", [False, True])
def test_renumbered_batch_gives_the_same_points_loss_and_gradients(deferred):
    """The U-Net step on a Z-order-renumbered batch against the same batch in the reference's numbering, fp32: per-point scores
    (max-abs error over max-abs value 1e-4), loss, and every parameter gradient (a sum over rows in another order: 2e-3 of its
    norm on the module-by-module path; 5e-3 with the deferred weight gradients on, where levels 4-7 run as one extension call
    whose BatchNorm statistics are fp64 totals of per-workgroup fp32 sums — the workgroups' rows change with the numbering —
    and the gradient of a BatchNorm weight is a cancelling sum over 30 k rows)."""
    from doda_amd.collate import reorder_voxels
    from doda_amd.model import SparseConvNet, cross_entropy, default_cfg, voxelize_and_run
    from doda_amd.scene import make_batch
    from doda_amd.spconv import functional as Fsp
    from tests.util import deterministic_init
    d = torch.device("cuda:0")
    cfg = default_cfg()
    base = make_batch(2, 60000, 23)
    got = []
    if deferred and not Fsp.set_deferred_wgrad(True):
        pytest.skip("compiled extension not built")
    if not deferred:
        Fsp.set_deferred_wgrad(False)
    try:
        for order in ("first", "morton"):
            b = reorder_voxels(base, order)
            bd = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
            net = deterministic_init(SparseConvNet(cfg), seed=4).to(d).train()
            scores = voxelize_and_run(cfg, net, bd, d, feature_dtype=torch.float32)
            loss = cross_entropy(scores, bd["labels"], ignore_index=255)
            loss.backward()
            torch.cuda.synchronize()
            got.append((scores.detach().float().cpu(), float(loss), {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}))
    finally:
        Fsp.set_deferred_wgrad(False)
    (s0, l0, g0), (s1, l1, g1) = got
    assert float((s0 - s1).abs().max()) < 1e-4 * float(s0.abs().max())
    assert abs(l0 - l1) < 1e-5 * abs(l0)
    tol = 5e-3 if deferred else 2e-3
    for k, a in g0.items():
        assert float((a - g1[k]).norm()) <= tol * float(a.norm()) + 1e-7, k


@pytest.mark.gpu
def test_choose_voxel_order_follows_the_tile_overflow():
    """collate.choose_voxel_order: the 2 cm bench scene fits the tile lists in the reference's numbering ('first'); a 1 cm scene
    overflows most level-1 tiles ('morton'), and after the renumbering it fits."""
    from doda_amd.collate import choose_voxel_order, reorder_voxels
    from doda_amd.scene import make_batch
    from doda_amd.spconv import ops as sops
    d = torch.device("cuda:0")
    assert choose_voxel_order(make_batch(1, 150000, 1000), d) == "first"
    b = make_batch(1, 400000, 1000, voxel_scale=100)       # (18 % of its level-1 tiles above the list capacity)
    assert choose_voxel_order(b, d) == "morton"
    r = reorder_voxels(b, "morton")
    assert choose_voxel_order(r, d) == "first"
    idx = r["voxel_locs"].int().to(d)
    _, nt, _, over = sops._ext.build_pyramid_probe(idx, [int(v) for v in r["spatial_shape"]], 1, 1, -1, sops.TILE_MIN_ROWS, 1)
    assert nt > 1000 and over == 0
