"""Synthetic ScanNet-shaped scenes and the batch contract of DODA's collate.

Generator (SURVEY §8d / BASELINE.md §3): a box room (floor + 4 walls) with axis-aligned furniture
boxes (5 faces each), points on a dense regular lattice per surface thinned to ~1.3 points per voxel
(mesh-vertex-like, see _sample / make_scene), centred, random rotation about z, +-5 mm jitter, then scaled by
`voxel_scale` (50 = 2 cm), shifted to the positive octant and truncated to integers — the same coordinate
pipeline as reference dataset/scannet.py:76-78.  The room size is searched so the scene has `target_voxels`
active voxels (+-2 %); N ~= pts_per_voxel * M; 11.4-12.1 occupied 27-neighbours per voxel and a 4.0-4.5x
population drop per stride-2 level, as the SURVEY 8d probe measured.
Scene s of a batch seeded `seed` uses seed + s.

make_batch() reproduces the dictionary of reference dataset/dataset.py:121-187 (collate_fn):
locs, voxel_locs, p2v_map, v2p_map, locs_float, feats (= centred xyz, dataset.py:142), labels,
offsets, spatial_shape (= clip(max+1, full_scale[0])), id.
"""
import numpy as np
import torch

from . import ops

_BASE_ROOM = (3.2, 4.0, 2.4)
_N_BOX = 5


def _surfaces(rng, scale):
    w, l, h = (d * scale for d in _BASE_ROOM)
    # (origin, edge u, edge v, label)
    surf = [((-w / 2, -l / 2, 0), (w, 0, 0), (0, l, 0), 1),          # floor
            ((-w / 2, -l / 2, 0), (w, 0, 0), (0, 0, h), 0),          # walls
            ((-w / 2, l / 2, 0), (w, 0, 0), (0, 0, h), 0),
            ((-w / 2, -l / 2, 0), (0, l, 0), (0, 0, h), 0),
            ((w / 2, -l / 2, 0), (0, l, 0), (0, 0, h), 0)]
    for b in range(_N_BOX):
        bw, bl, bh = rng.uniform(0.4, 1.2) * scale, rng.uniform(0.4, 1.2) * scale, rng.uniform(0.3, 0.9) * scale
        cx = rng.uniform(-w / 2 + bw / 2, w / 2 - bw / 2)
        cy = rng.uniform(-l / 2 + bl / 2, l / 2 - bl / 2)
        x0, y0, lab = cx - bw / 2, cy - bl / 2, 2 + (b % 18)
        surf += [((x0, y0, bh), (bw, 0, 0), (0, bl, 0), lab),        # top
                 ((x0, y0, 0), (bw, 0, 0), (0, 0, bh), lab),
                 ((x0, y0 + bl, 0), (bw, 0, 0), (0, 0, bh), lab),
                 ((x0, y0, 0), (0, bl, 0), (0, 0, bh), lab),
                 ((x0 + bw, y0, 0), (0, bl, 0), (0, 0, bh), lab)]
    return surf


OVERSAMPLE = 3.0   # lattice density of the surface sampling relative to the points that are kept (see make_scene)


def _sample(seed, scale, voxel_scale, pts_per_voxel):
    """Mesh-vertex-like sampling: every surface carries a regular lattice of points in raster order, dense
    enough (pitch 1/(voxel_scale*sqrt(OVERSAMPLE*pts_per_voxel)): ~1 cm at 2 cm voxels) that every voxel a
    surface passes through is hit, including the ones it only clips — those are what give a scanned surface its
    11-12 occupied 27-neighbours per voxel (SURVEY 8d probe).  make_scene thins the points back to
    `pts_per_voxel` per voxel without emptying any voxel.  (A lattice at 1.3 points per voxel directly, the
    round-1/2 generator, left the clipped voxels empty: 9.7 neighbours per voxel; i.i.d. uniform samples at the
    same N leave half of the surface voxels empty.)"""
    rng = np.random.default_rng(seed)
    surf = _surfaces(rng, scale)
    pitch = 1.0 / (voxel_scale * np.sqrt(pts_per_voxel * OVERSAMPLE))
    pts, labels = [], []
    for (o, u, v, lab) in surf:
        lu, lv = np.linalg.norm(u), np.linalg.norm(v)
        nu, nv = max(int(lu / pitch), 1), max(int(lv / pitch), 1)
        a = (np.arange(nu) + 0.5) / nu
        b = (np.arange(nv) + 0.5) / nv
        aa, bb = np.meshgrid(a, b, indexing="ij")
        p = np.asarray(o) + aa.reshape(-1, 1) * np.asarray(u) + bb.reshape(-1, 1) * np.asarray(v)
        pts.append(p)
        labels.append(np.full(p.shape[0], lab, dtype=np.int64))
    xyz = np.concatenate(pts, 0)
    labels = np.concatenate(labels, 0)
    xyz -= xyz.mean(0)
    th = rng.uniform(0, 2 * np.pi)
    rot = np.array([[np.cos(th), np.sin(th), 0], [-np.sin(th), np.cos(th), 0], [0, 0, 1]])
    xyz = xyz @ rot
    xyz += rng.uniform(-0.005, 0.005, size=xyz.shape)
    return xyz.astype(np.float32), labels


def _count_voxels(xyz, voxel_scale):
    q = np.floor(xyz * voxel_scale - (xyz * voxel_scale).min(0)).astype(np.int64)
    key = (q[:, 0] * 4096 + q[:, 1]) * 4096 + q[:, 2]
    return np.unique(key).size


def make_scene(seed, target_voxels=150000, voxel_scale=50, pts_per_voxel=1.3):
    """-> (xyz int64 [N,3] voxel coords, xyz_mid float32 [N,3] metres centred, labels int64 [N])."""
    # the voxel count of a surface scene grows with the square of the room scale: a secant step on sqrt(M)
    # lands within 2 % in two or three samplings (bisection needed ~12, each over a few million lattice points)
    scale = float(np.sqrt(target_voxels / 190000.0)) * 50.0 / voxel_scale
    lo, hi = 0.0, float("inf")
    for _ in range(20):
        xyz_mid, labels = _sample(seed, scale, voxel_scale, pts_per_voxel)
        m = _count_voxels(xyz_mid, voxel_scale)
        if abs(m - target_voxels) <= 0.02 * target_voxels:
            break
        lo, hi = (max(lo, scale), hi) if m < target_voxels else (lo, min(hi, scale))
        nxt = scale * float(np.sqrt(target_voxels / max(m, 1)))
        if not (lo < nxt < hi):   # the secant step left the bracket (tiny scenes: the count is not smooth)
            nxt = 0.5 * (lo + hi) if np.isfinite(hi) else 2.0 * scale
        scale = nxt
    xyz = xyz_mid.astype(np.float64) * voxel_scale
    xyz -= xyz.min(0)
    xyz = xyz.astype(np.int64)
    # thin the dense lattice to ~pts_per_voxel points per voxel (ScanNet-like multiplicity) keeping
    # every voxel occupied: the first point of each voxel stays, the others with one common probability;
    # raster order is preserved
    key = (xyz[:, 0] * 4096 + xyz[:, 1]) * 4096 + xyz[:, 2]
    _, first = np.unique(key, return_index=True)
    keep = np.zeros(xyz.shape[0], dtype=bool)
    keep[first] = True
    extra = max(0.0, (pts_per_voxel - 1.0) * first.size) / max(xyz.shape[0] - first.size, 1)
    keep |= np.random.default_rng(seed ^ 0x5eed).random(xyz.shape[0]) < extra
    return xyz[keep], xyz_mid[keep], labels[keep]


def make_batch(n_scenes=4, target_voxels=150000, seed=1000, voxel_scale=50, full_scale=(128, 512),
               voxel_mode=4, pts_per_voxel=1.3):
    """Collate `n_scenes` synthetic scenes into DODA's batch dictionary (CPU tensors)."""
    locs, locs_float, labels, offsets = [], [], [], [0]
    for s in range(n_scenes):
        xyz, xyz_mid, lab = make_scene(seed + s, target_voxels, voxel_scale, pts_per_voxel)
        offsets.append(offsets[-1] + xyz.shape[0])
        locs.append(torch.cat([torch.full((xyz.shape[0], 1), s, dtype=torch.int64),
                               torch.from_numpy(xyz)], 1))
        locs_float.append(torch.from_numpy(xyz_mid))
        labels.append(torch.from_numpy(lab))
    locs = torch.cat(locs, 0)
    locs_float = torch.cat(locs_float, 0).to(torch.float32)
    labels = torch.cat(labels, 0).long()
    spatial_shape = np.clip((locs.max(0)[0][1:] + 1).numpy(), full_scale[0], None)
    voxel_locs, p2v_map, v2p_map = ops.voxelize_idx_host(locs, n_scenes, voxel_mode)
    # (v2p_map_t: the voxel -> point table transposed to the gather-table form [maxActive, M] the fused head's backward
    # reads — a property of the batch, produced by the loader next to the maps themselves)
    return {"locs": locs, "voxel_locs": voxel_locs, "p2v_map": p2v_map, "v2p_map": v2p_map,
            "v2p_map_t": v2p_map[:, 1:].t().contiguous(),
            "locs_float": locs_float, "feats": locs_float.clone(), "labels": labels,
            "offsets": torch.tensor(offsets, dtype=torch.int32), "spatial_shape": spatial_shape,
            "id": list(range(n_scenes))}
