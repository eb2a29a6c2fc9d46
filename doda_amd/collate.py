"""Device-resident collate: the batch dictionary of reference
dataset/dataset.py:121-187 (`collate_fn`) assembled on the GPU.

The reference concatenates the scenes and voxelises them on CPU inside DataLoader workers (~1 M
points/s per worker), then copies everything to the GPU.  Here the per-scene arrays are uploaded once
and the concatenation, the batch-index column and `voxelization_idx` (doda_voxelize_idx_assign/_fill:
hash build + first-touch numbering, bit-identical to the host path) run on the device; the only
host round trips are the output sizes of the voxelisation and the three maxima for `spatial_shape`.
Same keys, dtypes and shapes as the reference dictionary; tensors live on `device`."""
import numpy as np
import torch

from . import pointgroup_ops


def _dev(a, device, dtype=None):
    t = torch.from_numpy(a) if isinstance(a, np.ndarray) else a
    t = t.to(device, non_blocking=True)
    return t if dtype is None else t.to(dtype)


def collate_device(items, device, batch_size=None, voxel_mode=4, full_scale=(128, 512)):
    """items: list of (xyz int [n,3] voxel coordinates, xyz_mid float [n,3], label [n], idx, *others)
    exactly as a DODA dataset's __getitem__ returns them (dataset/dataset.py:136).  `others[0]` may carry
    selected_idx / mask1 / mask2 arrays (concatenated) and mix_idx / tar_tail_splits /
    tar_splits_class_ratio (collected), as in the reference."""
    device = torch.device(device)
    if batch_size is None:
        batch_size = len(items)
    locs, locs_float, labels, ids, offsets = [], [], [], [], [0]
    extra_cat = {"selected_idx": [], "mask1": [], "mask2": []}
    mix_idx, tar_tail_splits, tar_ratio = [], [], []
    for i, item in enumerate(items):
        xyz, xyz_mid, label, idx, *others = item
        n = xyz.shape[0]
        offsets.append(offsets[-1] + n)
        xyz_d = _dev(xyz, device, torch.int64)
        locs.append(torch.cat([torch.full((n, 1), i, dtype=torch.int64, device=device), xyz_d], 1))
        locs_float.append(_dev(xyz_mid, device, torch.float32))
        labels.append(_dev(label, device, torch.int64))
        ids.append(idx)
        if others:
            o = others[0]
            for key in extra_cat:
                if key in o:
                    extra_cat[key].append(_dev(o[key], device))
            if "mix_idx" in o:
                mix_idx.append(o["mix_idx"])
            if "tar_tail_splits" in o:
                tar_tail_splits.extend(o["tar_tail_splits"])
            if "tar_splits_class_ratio" in o:
                tar_ratio.append(o["tar_splits_class_ratio"])
    locs = torch.cat(locs, 0).contiguous()
    locs_float = torch.cat(locs_float, 0)
    labels = torch.cat(labels, 0)
    if locs.shape[0] > 0:
        top = (locs[:, 1:].max(0)[0] + 1).cpu().numpy()
        spatial_shape = np.clip(top, full_scale[0], None)
    else:
        spatial_shape = np.array([full_scale[0]] * 3, dtype=np.int64)
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(locs, batch_size, voxel_mode)
    out = {"locs": locs, "voxel_locs": voxel_locs, "p2v_map": p2v_map, "v2p_map": v2p_map,
           "v2p_map_t": v2p_map[:, 1:].t().contiguous(),   # gather-table form for the fused head's backward (doda_amd.model)
           "locs_float": locs_float, "feats": locs_float.clone(), "labels": labels,
           "offsets": torch.tensor(offsets, dtype=torch.int32), "spatial_shape": spatial_shape, "id": ids,
           "mix_idx": mix_idx, "tar_tail_splits": tar_tail_splits}
    for key, parts in extra_cat.items():
        out[key] = torch.cat(parts, 0) if parts else []
    ratio = []
    for r in tar_ratio:
        ratio = ratio + r
    out["tar_splits_class_ratio"] = ratio
    return out


def collate_device_concat(hb, device, voxel_mode=4):
    """The batch dictionary from a host-side concatenation (doda_amd.loader.host_collate, run in the DataLoader workers like
    the reference's collate_fn): three uploads, the widening to the reference's dtypes and `voxelization_idx` on the device.
    Same keys and values as collate_device on the same scenes."""
    device = torch.device(device)
    locs = hb["locs32"].to(device, non_blocking=True).to(torch.int64)
    locs_float = hb["locs_float"].to(device, non_blocking=True)
    labels = hb["labels32"].to(device, non_blocking=True).to(torch.int64)
    batch_size = hb["offsets"].numel() - 1
    voxel_locs, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(locs, batch_size, voxel_mode)
    return {"locs": locs, "voxel_locs": voxel_locs, "p2v_map": p2v_map, "v2p_map": v2p_map,
            "v2p_map_t": v2p_map[:, 1:].t().contiguous(),
            "locs_float": locs_float, "feats": locs_float.clone(), "labels": labels,
            "offsets": hb["offsets"], "spatial_shape": hb["spatial_shape"], "id": hb["id"],
            "mix_idx": [], "tar_tail_splits": [], "selected_idx": [], "mask1": [], "mask2": [], "tar_splits_class_ratio": []}


# ---------------------------------------------------------------------------------------------------------------------
# Voxel order (round 5).  The reference numbers the voxels of a batch in the order their first point appears
# (lib/pointgroup_ops/src/voxelize/voxelize.cpp:61-120: the hash map's insertion order), i.e. in whatever order the scan's
# points are stored.  Nothing downstream depends on that numbering: the U-Net is equivariant under a permutation of the voxel
# rows, BatchNorm statistics and weight gradients are sums over rows, and the only things that name a voxel row are the two
# maps of the batch itself (p2v_map: point -> voxel, v2p_map: voxel -> points).  The LDS-staged kernels, on the other hand, live
# on the locality of the numbering: a tile of 256 consecutive rows stages every DISTINCT neighbour row once (tilebook.hpp), and
# how many there are is set by the shape of the region those 256 voxels cover — a strip of a raster-like scan has ~2.2 x 256,
# at 1 cm voxels more than the 1023 a tile can list, a compact patch ~1.4 x 256.  reorder_voxels() renumbers
# the voxels of every scene along a Z-order (Morton) curve of their coordinates and rewrites the two maps; per-point outputs,
# the loss and every parameter gradient are those of the reference's numbering up to the order of floating-point sums.
def _spread3(v):
    """bits of v (16 used) moved to every third position"""
    v = v & 0xFFFF
    v = (v | (v << 16)) & 0x0000FF0000FF
    v = (v | (v << 8)) & 0x00F00F00F00F
    v = (v | (v << 4)) & 0x0C30C30C30C3
    v = (v | (v << 2)) & 0x249249249249
    return v


def morton_keys(voxel_locs):
    """int64 sort keys of [M, 4] (batch, x, y, z) rows: batch index above the interleaved bits of the coordinates."""
    c = voxel_locs.long()
    if c.numel() and int(c[:, 1:].max()) >= 65536:
        raise ValueError("morton_keys: coordinates of 16 bits at most")
    return (c[:, 0] << 48) | (_spread3(c[:, 1]) << 2) | (_spread3(c[:, 2]) << 1) | _spread3(c[:, 3])


def reorder_voxels(batch, order="morton"):
    """The batch dictionary with its voxels renumbered (a shallow copy; `order`: "morton", or "first" = unchanged).
    Rewrites voxel_locs, v2p_map, v2p_map_t (rows permuted) and p2v_map (values renamed); works on CPU or device tensors."""
    if order in (None, "first") or batch["voxel_locs"].shape[0] == 0:
        return batch
    if order != "morton":
        raise ValueError("reorder_voxels: order is 'first' or 'morton'")
    out = dict(batch)
    perm = torch.argsort(morton_keys(batch["voxel_locs"]), stable=True)       # new row j holds old row perm[j]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    out["voxel_locs"] = batch["voxel_locs"][perm].contiguous()
    out["v2p_map"] = batch["v2p_map"][perm].contiguous()
    if batch.get("v2p_map_t") is not None:
        out["v2p_map_t"] = batch["v2p_map_t"][:, perm].contiguous()
    out["p2v_map"] = inv[batch["p2v_map"].long()].to(batch["p2v_map"].dtype)
    out["voxel_order"] = order
    return out


# share of level-1 tiles above the list capacity from which the renumbering is applied: such tiles run from the dense table INSIDE
# the tile kernel (slower than either path); the bench scene at 2 cm has 0.4 %, 1 cm scenes of 400 k / 500 k voxels 18 % / 69 %
TILE_OVERFLOW_SWITCH = 0.05


def choose_voxel_order(batch, device=None):
    """"first" or "morton" for batches like this one: builds the level-1 SubM rulebook of `batch` with its tilebook once and
    reads the builder's overflow counters (one device synchronisation — loaders call it for their first batch and keep the
    answer).  Measured on MI355X: at 2 cm the reference's numbering fits the tiles (0.4 % above the list) and
    the whole step is as fast in either order — conv_tile16 gains 5-9 % from the shorter lists, the dense-table gathers lose
    their runs of consecutive rows —, at 1 cm 74 % of the level-1 tiles overflow and the renumbered step is 30 % faster."""
    from .spconv import ops as sops
    vl = batch["voxel_locs"]
    if sops._ext is None or vl.shape[0] < sops.TILE_MIN_ROWS or not sops.TILE_KERNEL:
        return "first"
    dev = torch.device(device) if device is not None else vl.device
    if dev.type != "cuda":
        return "first"
    idx = vl.to(dev).int().contiguous()
    shape = [int(v) for v in batch["spatial_shape"]]
    nb = int(batch["offsets"].numel() - 1)
    _, nt, _, over = sops._ext.build_pyramid_probe(idx, shape, nb, 1, -1, sops.TILE_MIN_ROWS, 1)
    return "morton" if (nt > 0 and over > TILE_OVERFLOW_SWITCH * nt) else "first"
