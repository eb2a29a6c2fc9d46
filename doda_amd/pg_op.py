"""`PG_OP`-shaped module: the functions DODA's lib/pointgroup_ops/functions/pointgroup_ops.py
calls on its pybind extension (reference lib/pointgroup_ops/src/pointgroup_ops_api.cpp:6-26),
with the same positional signatures, backed by libdoda_hip.so.  Out of scope (PointGroup
instance-segmentation leftovers with no call site in DODA): bfs_cluster, roipool_*,
get_iou, sec_*."""
from . import ops as _ops


def voxelize_idx(coords, output_coords, input_map, output_map, batch_size, mode):
    """pointgroup_ops.cpp:14-17 -> voxelize.cpp:10-31.  Resizes output_coords / output_map like
    the reference (`resize_` + fill).  CPU tensors run the fork-safe host path (DataLoader
    workers); device tensors run the HIP path."""
    if coords.is_cuda:
        oc, im, om = _ops.voxelize_idx_device(coords, batch_size, mode)
    else:
        oc, im, om = _ops.voxelize_idx_host(coords, batch_size, mode)
    output_coords.resize_(oc.shape).copy_(oc)
    output_map.resize_(om.shape).copy_(om)
    input_map.copy_(im)


def voxelize_fp(feats, output_feats, output_map, mode, n_active, max_active, n_plane):
    _ops.voxelize_fp(feats, output_feats, output_map, mode, n_active, max_active, n_plane)


def voxelize_bp(d_output_feats, d_feats, output_map, mode, n_active, max_active, n_plane):
    _ops.voxelize_bp(d_output_feats, d_feats, output_map, mode, n_active, max_active, n_plane)


def point_recover_fp(feats, output_feats, idx_map, n_active, max_active, n_plane):
    _ops.point_recover_fp(feats, output_feats, idx_map, n_active, max_active, n_plane)


def point_recover_bp(d_output_feats, d_feats, idx_map, n_active, max_active, n_plane):
    _ops.point_recover_bp(d_output_feats, d_feats, idx_map, n_active, max_active, n_plane)


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, mean_active, radius):
    return _ops.ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, mean_active,
                                  radius)


def knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, idx, n, m, k):
    _ops.knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, idx, n, m, k)


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            "PG_OP.%s is outside doda_amd's scope: it has no call site in DODA "
            "(PointGroup instance-segmentation leftover)" % name)
    fn.__name__ = name
    return fn


for _n in ("bfs_cluster", "roipool_fp", "roipool_bp", "get_iou", "sec_mean", "sec_mean_bp",
           "sec_min", "sec_max"):
    globals()[_n] = _out_of_scope(_n)
