"""Host-side mirror of DODA's pointops2 wrapper for the one entry point DODA calls
(reference lib/pointops2/functions/pointops2.py:54-71, call site model/unet.py:135-138)."""
import torch
from torch.autograd import Function

from . import pointops2_cuda as pointops_cuda


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz, offset, new_offset):
        """xyz (n,3), new_xyz (m,3) float32; offset, new_offset (b+1) int32 with a leading 0.
        Returns idx (m,nsample) int32 and dist (m,nsample) = sqrt(dist2), ascending."""
        if new_xyz is None:
            new_xyz = xyz
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        m = new_xyz.shape[0]
        idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
        dist2 = torch.zeros((m, nsample), dtype=torch.float32, device=xyz.device)
        pointops_cuda.knnquery_cuda(m, nsample, xyz, new_xyz, offset[1:], new_offset[1:], idx, dist2)
        return idx, torch.sqrt(dist2)


knnquery = KNNQuery.apply
