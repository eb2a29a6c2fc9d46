"""The five entry points DODA imports from `lib.pointgroup_ops.functions.pointgroup_ops`
(reference lib/pointgroup_ops/functions/pointgroup_ops.py:13-153), as plain functions over doda_amd.ops.

    voxelization_idx(coords, batchsize, mode=4)  -> (voxel_coords int64 [M,ncol], p2v int32 [N], v2p int32 [M,1+maxActive])
    voxelization(feats, map_rule, mode=4)         -> [M,C]      differentiable w.r.t. feats
    point_recover(feats, map_rule, nPoint)        -> [nPoint,C] differentiable w.r.t. feats
    ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive) -> (idx int32 [total], start_len int32 [n,2])
    knn / knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, k) -> idx int32 [n,k]
plus `.apply` aliases under the reference's class names (Voxelization_Idx, Voxelization, PointRecover, BallQueryBatchP,
KNNBatch).

Names, positional argument order, dtypes and return tuples are the reference's (its callers: dataset/dataset.py:182,
model/unet.py:88, model/unet.py:136-141); the bodies are this repository's.  The two differentiable ops are ONE
autograd function: pooling rows of `feats` through a voxel->point map and spreading voxel rows back to points are
transposes of each other, so each one's backward is the other's forward kernel.
"""
import torch
from torch.autograd import Function

from . import ops as _ops

# kernel pairs (forward, backward) per direction; all four take (src, dst_zeroed, map, [mode,] M, maxActive, C)
_POOL, _SPREAD = 0, 1


def _launch(direction, backward, src, dst, rules, mode):
    m, width = rules.shape
    c = src.shape[1]
    if direction == _POOL:
        (_ops.voxelize_bp if backward else _ops.voxelize_fp)(src, dst, rules, mode, m, width - 1, c)
    else:
        (_ops.point_recover_bp if backward else _ops.point_recover_fp)(src, dst, rules, m, width - 1, c)


class _MapRows(Function):
    """dst = A(src) with A the row map of `rules` (pool: points -> voxels with mode 3 sum / 4 mean; spread:
    voxels -> points); backward applies A^T through the partner kernel."""

    @staticmethod
    def forward(ctx, src, rules, direction, mode, n_dst):
        if not (src.is_contiguous() and rules.is_contiguous()):
            raise AssertionError("pointgroup_ops: feats and map_rule must be contiguous")   # reference asserts
        dst = src.new_zeros((n_dst, src.shape[1]), dtype=torch.float32)
        _launch(direction, False, src, dst, rules, mode)
        ctx.rules, ctx.key, ctx.n_src = rules, (direction, mode), src.shape[0]
        return dst

    @staticmethod
    def backward(ctx, d_dst):
        direction, mode = ctx.key
        d_src = d_dst.new_zeros((ctx.n_src, d_dst.shape[1]), dtype=torch.float32)
        _launch(direction, True, d_dst.contiguous(), d_src, ctx.rules, mode)
        return d_src, None, None, None, None


def voxelization(feats, map_rule, mode=4):
    """Voxel features = sum (mode 3) / mean (mode 4) of the voxel's points; map_rule = v2p_map."""
    return _MapRows.apply(feats, map_rule, _POOL, mode, map_rule.shape[0])


def point_recover(feats, map_rule, nPoint):
    """Voxel features copied back to the voxel's points."""
    return _MapRows.apply(feats, map_rule, _SPREAD, 4, int(nPoint))


@torch.no_grad()
def voxelization_idx(coords, batchsize, mode=4):
    """coords int64 [N, 3|4] (batch index first when 4 columns).  CPU tensors take the fork-safe host path (this is
    what DataLoader workers call), device tensors the HIP path."""
    if not coords.is_contiguous():
        raise AssertionError("voxelization_idx: coords must be contiguous")
    build = _ops.voxelize_idx_device if coords.is_cuda else _ops.voxelize_idx_host
    voxel_coords, p2v, v2p = build(coords, batchsize, mode)
    return voxel_coords, p2v, v2p


@torch.no_grad()
def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive):
    """Neighbours within `radius` inside each batch item, packed; start_len[i] = (offset, count).  The packed
    buffer holds n * meanActive entries; when the kernel reports more, it is re-sized to fit and the query repeated
    (the reference's contract: pointgroup_ops.py:137-144)."""
    for t in (coords, batch_idxs, batch_offsets):
        if not (t.is_cuda and t.is_contiguous()):
            raise AssertionError("ballquery_batch_p: device-resident contiguous tensors expected")
    n = coords.shape[0]
    if n == 0:
        return (torch.zeros(0, dtype=torch.int32, device=coords.device),
                torch.zeros((0, 2), dtype=torch.int32, device=coords.device))
    cap = int(meanActive)
    total = None
    while total is None or total > n * cap:
        if total is not None:
            cap = total // n + 1
        idx = torch.zeros(n * cap, dtype=torch.int32, device=coords.device)
        start_len = torch.zeros((n, 2), dtype=torch.int32, device=coords.device)
        total = int(_ops.ballquery_batch_p(coords, batch_idxs, batch_offsets, idx, start_len, n, cap, radius))
    return idx[:total], start_len


@torch.no_grad()
def knn(xyz, query_xyz, batch_idxs, query_batch_offsets, k):
    """For every point of `xyz`: its k nearest points of `query_xyz` in the same batch item (k <= 40)."""
    for t in (xyz, query_xyz, batch_idxs, query_batch_offsets):
        if not (t.is_cuda and t.is_contiguous()):
            raise AssertionError("knn: device-resident contiguous tensors expected")
    n, m = xyz.shape[0], query_xyz.shape[0]
    idx = torch.zeros((n, k), dtype=torch.int32, device=xyz.device)
    _ops.knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, idx, n, m, k)
    return idx


knn_batch = knn   # the reference's name for it (pointgroup_ops.py:380)


class _ApplyAlias:
    """`Name.apply(...)` for code written against the reference's Function classes (pointgroup_ops.py:13,44,80,117,349):
    the same plain functions behind the class names."""

    def __init__(self, fn):
        self.apply = fn

    def __call__(self, *args, **kwargs):
        return self.apply(*args, **kwargs)


Voxelization_Idx = _ApplyAlias(voxelization_idx)
Voxelization = _ApplyAlias(voxelization)
PointRecover = _ApplyAlias(point_recover)
BallQueryBatchP = _ApplyAlias(ballquery_batch_p)
KNNBatch = _ApplyAlias(knn)
