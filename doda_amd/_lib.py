"""ctypes binding of libdoda_hip.so (include/doda_hip.h).

There is no fallback: if the shared library is missing or lacks a symbol, importing the product
path raises.  PyTorch is used only for device memory and streams; every signature below is
plain pointers + sizes.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdoda_hip.so")

c_i32, c_i64, c_f32, c_sz, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_void_p

# name -> (restype, argtypes); mirrors include/doda_hip.h one to one
_SIGNATURES = {
    "doda_abi_version": (c_i32, []),
    "doda_strerror": (C.c_char_p, [c_i32]),
    "doda_voxelize_idx_h": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, C.POINTER(c_vp),
                                    C.POINTER(c_i32), C.POINTER(c_i32)]),
    "doda_voxelize_idx_fill_h": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "doda_voxelize_idx_free_h": (None, [c_vp]),
    "doda_voxelize_idx_workspace_bytes": (c_sz, [c_i32]),
    "doda_voxelize_idx_assign": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "doda_voxelize_idx_fill": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                       c_sz, c_vp]),
    "doda_voxelize_fp": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "doda_voxelize_bp": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "doda_voxelize_fp_rows": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "doda_point_recover_fp": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "doda_point_recover_bp": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "doda_rulebook_workspace_bytes": (c_sz, [c_i32]),
    "doda_rulebook_subm": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "doda_rulebook_down2_assign": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_sz, c_vp]),
    "doda_rulebook_down2_tables": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "doda_rulebook_conv_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "doda_rulebook_conv_assign": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                          c_sz, c_vp]),
    "doda_rulebook_conv_tables": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp,
                                          c_i32, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "doda_rulebook_subm_generic": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "doda_rulebook_pairs_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "doda_rulebook_pairs": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_sz,
                                    c_vp]),
    "doda_spconv_gather_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "doda_spconv_pack_desc_bytes": (c_sz, []),
    "doda_spconv_pack_plan_h": (c_i32, [c_vp, c_i32, c_vp, C.POINTER(c_i32)]),
    "doda_spconv_pack_multi": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "doda_rulebook_pairs_tile": (c_i32, []),
    "doda_spconv_wgrad_multi_workspace_bytes": (c_sz, [c_vp, c_i32]),
    "doda_spconv_wgrad_multi_desc_bytes": (c_sz, [c_i32]),
    "doda_spconv_wgrad_multi": (c_i32, [c_vp, c_i32, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "doda_maxpool_fwd_f32": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "doda_maxpool_bwd_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "doda_cross_entropy_workspace_bytes": (c_sz, [c_i32]),
    "doda_seg_meters": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, C.c_int64, c_vp, c_vp]),
    "doda_cross_entropy_fwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, C.c_int64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "doda_cross_entropy_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, C.c_int64, c_vp, c_vp]),
    "doda_tilebook_tile": (c_i32, []),
    "doda_tilebook_umax": (c_i32, []),
    "doda_tilebook_bytes": (c_sz, [c_i32, c_i32]),
    "doda_tilebook_build": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "doda_set_option": (c_i32, [c_i32, c_i32]),
    "doda_get_option": (c_i32, [c_i32]),
    "doda_spconv_stats_capacity": (c_sz, [c_i32]),
    "doda_spconv_gather_ex": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32,
                                      c_i32, c_vp, c_sz, c_vp, c_vp]),
    "doda_bn_relu_fwd_stats": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "doda_bn_relu_bwd_stats": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32,
                                          c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "doda_bn_relu_fwd_totals": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "doda_bn_relu_bwd_totals": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32,
                                        c_vp, c_vp, c_vp, c_vp]),
    "doda_bn_relu_bwd_add": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp,
                                        c_vp, c_vp, c_vp, c_sz, c_vp]),
    "doda_bn_relu_apply": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "doda_bn_fwd_final": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "doda_cast_colsum_blocks": (c_i32, [C.c_int64, c_i32]),
    "doda_cast_colsum_f32_bf16": (c_i32, [c_vp, C.c_int64, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "doda_pad_channels": (c_i32, [c_vp, C.c_int64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "doda_sgd_multi_desc_bytes": (c_sz, [c_i32]),
    "doda_sgd_multi": (c_i32, [c_vp, c_i32, C.c_double, C.c_double, C.c_double, C.c_double, c_i32, c_i32, c_vp,
                               c_sz, c_vp]),
    "doda_bn_workspace_bytes": (c_sz, [c_i32, c_i32]),
    "doda_bn_relu_fwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "doda_bn_relu_bwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                 c_vp, c_vp, c_vp, c_sz, c_vp]),
    "doda_knnquery": (c_i32, [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "doda_knn_batch": (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "doda_ballquery_workspace_bytes": (c_sz, [c_i32]),
    "doda_ballquery_batch_p": (c_i32, [c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       C.POINTER(c_i32), c_vp, c_sz, c_vp]),
    "doda_layers_run": (c_i32, [c_vp, c_i32, c_i32, C.POINTER(c_i32), c_vp]),
    "doda_head_ce_blocks": (c_i32, [c_i32]),
    "doda_head_dw_blocks": (c_i32, [c_i32]),
    "doda_head_dw_bf16": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "doda_head_ce_fwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "doda_head_ce_bwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, C.c_int64, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, c_i32, c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
OPT_TILE_KERNEL, OPT_WLDS_KERNEL, OPT_WDMA_KERNEL, OPT_TILE_PIPELINE, OPT_TILE_DUAL, OPT_CONV_UP = 1, 2, 3, 4, 5, 6   # doda_set_option / doda_get_option
OPT_PRE_FWD_ROWS, OPT_PRE_BWD_ROWS = 7, 8   # (row thresholds of doda_layers_run's BatchNorm folding)
ABI_VERSION = 12  # include/doda_hip.h DODA_ABI_VERSION

_lib = None


class DodaNativeError(RuntimeError):
    pass


def lib():
    """The loaded C-ABI library; raises DodaNativeError if it is not built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DodaNativeError(
                "libdoda_hip.so is not built (%s). Run `python -m doda_amd.build` "
                "(hipcc, gfx950). doda_amd has no CPU fallback for its native ops." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise DodaNativeError("libdoda_hip.so lacks symbol %s (stale build?)" % name) from e
            fn.restype = res
            fn.argtypes = args
        if handle.doda_abi_version() != ABI_VERSION:
            raise DodaNativeError("libdoda_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().doda_strerror(status).decode()
        raise DodaNativeError("%s failed: %s (%d)" % (what, msg, status))
