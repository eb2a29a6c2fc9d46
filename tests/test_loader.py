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
