"""Host-side mirror of DODA's pointgroup_ops wrapper module
(reference lib/pointgroup_ops/functions/pointgroup_ops.py): same function names, argument order,
return values and error behaviour for the entry points on DODA's hot path, over doda_amd.pg_op.

    voxelization_idx(coords, batchsize, mode=4) -> (output_coords, input_map, output_map)
    voxelization(feats, map_rule, mode=4)       -> output_feats           (differentiable)
    point_recover(feats, map_rule, nPoint)      -> output_feats           (differentiable)
    ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, meanActive) -> (idx, start_len)
    knn(xyz, query_xyz, batch_idxs, query_batch_offsets, k) -> idx
"""
import torch
from torch.autograd import Function

from . import pg_op as PG_OP


class Voxelization_Idx(Function):
    """pointgroup_ops.py:13-42.  coords: int64 (N, 3|4), contiguous; CPU in DODA's collate
    (dataset/dataset.py:182), device tensors take the HIP path."""

    @staticmethod
    def forward(ctx, coords, batchsize, mode=4):
        assert coords.is_contiguous()
        n = coords.size(0)
        output_coords = coords.new_empty(0)
        input_map = torch.zeros(n, dtype=torch.int32, device=coords.device)
        output_map = input_map.new_empty(0)
        PG_OP.voxelize_idx(coords, output_coords, input_map, output_map, batchsize, mode)
        return output_coords, input_map, output_map

    @staticmethod
    def backward(ctx, a=None, b=None, c=None):
        return None


voxelization_idx = Voxelization_Idx.apply


class Voxelization(Function):
    """pointgroup_ops.py:44-77: feats (N,C) float32 device, map_rule (M,1+maxActive) int32."""

    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        assert map_rule.is_contiguous()
        assert feats.is_contiguous()
        n, c = feats.size()
        m = map_rule.size(0)
        max_active = map_rule.size(1) - 1
        output_feats = torch.zeros((m, c), dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, mode, max_active, n)
        PG_OP.voxelize_fp(feats, output_feats, map_rule, mode, m, max_active, c)
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, mode, max_active, n = ctx.for_backwards
        m, c = d_output_feats.size()
        d_feats = torch.zeros((n, c), dtype=torch.float32, device=d_output_feats.device)
        PG_OP.voxelize_bp(d_output_feats.contiguous(), d_feats, map_rule, mode, m, max_active, c)
        return d_feats, None, None


voxelization = Voxelization.apply


class PointRecover(Function):
    """pointgroup_ops.py:80-117: voxel feats (M,C) -> point feats (nPoint,C)."""

    @staticmethod
    def forward(ctx, feats, map_rule, nPoint):
        assert map_rule.is_contiguous()
        assert feats.is_contiguous()
        m, c = feats.size()
        max_active = map_rule.size(1) - 1
        output_feats = torch.zeros((nPoint, c), dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, max_active, m)
        PG_OP.point_recover_fp(feats, output_feats, map_rule, m, max_active, c)
        return output_feats

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, max_active, m = ctx.for_backwards
        n, c = d_output_feats.size()
        d_feats = torch.zeros((m, c), dtype=torch.float32, device=d_output_feats.device)
        PG_OP.point_recover_bp(d_output_feats.contiguous(), d_feats, map_rule, m, max_active, c)
        return d_feats, None, None


point_recover = PointRecover.apply


class BallQueryBatchP(Function):
    """pointgroup_ops.py:120-153, including the grow-and-retry loop on meanActive."""

    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, radius, meanActive):
        n = coords.size(0)
        assert coords.is_contiguous() and coords.is_cuda
        assert batch_idxs.is_contiguous() and batch_idxs.is_cuda
        assert batch_offsets.is_contiguous() and batch_offsets.is_cuda
        while True:
            idx = torch.zeros(n * meanActive, dtype=torch.int32, device=coords.device)
            start_len = torch.zeros((n, 2), dtype=torch.int32, device=coords.device)
            n_active = PG_OP.ballquery_batch_p(coords, batch_idxs, batch_offsets, idx, start_len, n,
                                               meanActive, radius)
            if n_active <= n * meanActive:
                break
            meanActive = int(n_active // n + 1)
        return idx[:n_active], start_len

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


ballquery_batch_p = BallQueryBatchP.apply


class KNN(Function):
    """k nearest points of query_xyz (same batch item) for every point of xyz; k <= 40."""

    @staticmethod
    def forward(ctx, xyz, query_xyz, batch_idxs, query_batch_offsets, k):
        n, m = xyz.size(0), query_xyz.size(0)
        assert xyz.is_contiguous() and xyz.is_cuda
        assert query_xyz.is_contiguous() and query_xyz.is_cuda
        assert batch_idxs.is_contiguous() and batch_idxs.is_cuda
        assert query_batch_offsets.is_contiguous() and query_batch_offsets.is_cuda
        idx = torch.zeros((n, k), dtype=torch.int32, device=xyz.device)
        PG_OP.knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, idx, n, m, k)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


knn = KNN.apply
