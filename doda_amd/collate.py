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
