"""`pointops.knnquery` as DODA calls it (reference lib/pointops2/functions/pointops2.py:54-71; call site
model/unet.py:135-138): a plain function over doda_amd.ops — the query returns indices and distances, nothing
differentiable, so no autograd class is involved."""
import torch

from . import ops as _ops


@torch.no_grad()
def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """For each row of new_xyz [m,3] (None: xyz itself) its `nsample` nearest rows of xyz [n,3] inside the same batch
    segment; offset / new_offset: int32 [B+1] segment ends with a leading 0 (stripped before the native call, as the
    reference does).  Returns (idx int32 [m,nsample] ascending by distance, dist float32 [m,nsample] = sqrt of the
    squared distance)."""
    queries = xyz if new_xyz is None else new_xyz
    if not (xyz.is_contiguous() and queries.is_contiguous()):
        raise AssertionError("knnquery: contiguous xyz / new_xyz expected")
    m = queries.shape[0]
    idx = xyz.new_zeros((m, nsample), dtype=torch.int32)
    dist2 = xyz.new_zeros((m, nsample), dtype=torch.float32)
    _ops.knnquery(m, nsample, xyz, queries, offset[1:], new_offset[1:], idx, dist2)
    return idx, dist2.sqrt_()


class KNNQuery:   # `KNNQuery.apply(...)`: the reference's class name for the same call
    apply = staticmethod(knnquery)
