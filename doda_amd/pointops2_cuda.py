"""`pointops2_cuda`-shaped module (reference lib/pointops2/src/pointops_api.cpp:12-23).  DODA
calls only knnquery_cuda (lib/pointops2/functions/pointops2.py:67); the Point-Transformer
operators (sampling, grouping, interpolation, subtraction, aggregation) have no call site in DODA
and are out of scope."""
from . import ops as _ops


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    """knnquery/knnquery_cuda.cpp:8-17: offset/new_offset are END offsets per batch item."""
    _ops.knnquery(m, nsample, xyz.contiguous(), new_xyz.contiguous(), offset.contiguous(),
                  new_offset.contiguous(), idx, dist2)


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("pointops2_cuda.%s is outside doda_amd's scope (no DODA call site)" % name)
    fn.__name__ = name
    return fn


for _n in ("furthestsampling_cuda", "furthestsampling_dim_cuda", "grouping_forward_cuda",
           "grouping_backward_cuda", "interpolation_forward_cuda", "interpolation_backward_cuda",
           "subtraction_forward_cuda", "subtraction_backward_cuda", "aggregation_forward_cuda",
           "aggregation_backward_cuda"):
    globals()[_n] = _out_of_scope(_n)
