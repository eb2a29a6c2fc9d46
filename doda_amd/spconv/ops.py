"""Rulebook construction at the level of spconv v1.2 `spconv.ops` (get_conv_output_size,
get_indice_pairs).  The geometries DODA instantiates — SubM with a cubic kernel of 1 or 3 and the
kernel-2 / stride-2 / padding-0 convolution (model/unet.py:36, model/unet_block.py:18-29,48,70,78) —
have dedicated native builders; any other kernel_size / stride / padding / dilation with a kernel
volume of at most 27 goes through the generic one (doda_rulebook_conv_*, doda_rulebook_subm_generic)."""
import os

import numpy as np

import torch

from .. import ops as _ops
from .._ext import ext as _ext
from .core import IndiceData

PAIRS_MIN_ROWS = 65536   # as doda_amd.spconv.functional.PAIRS_MIN_ROWS (pair lists only for large rulebooks)
WGRAD_PAIRS_DOWN = os.environ.get("DODA_WGRAD_PAIRS_DOWN", "0") == "1"   # (as doda_amd.spconv.functional: strided rulebooks' lists)


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        out = [int(x) for x in v]
        if len(out) != 3:
            raise ValueError("expected 3 values, got %r" % (v,))
        return out
    return [int(v)] * 3


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def build_subm(indices, batch_size, spatial_shape, ksize):
    k = _triple(ksize)
    if any(v % 2 == 0 or v < 1 for v in k) or k[0] * k[1] * k[2] > 27:
        raise NotImplementedError("doda_amd SubMConv3d supports odd kernel sizes with volume <= 27, got %r" % (k,))
    if k[0] == k[1] == k[2]:
        tbl = _ops.rulebook_subm(indices, spatial_shape, batch_size, k[0])
    else:
        tbl = _ops.rulebook_subm_generic(indices, spatial_shape, batch_size, k)
    return IndiceData("subm", indices, indices, list(spatial_shape), list(spatial_shape), tbl)


def build_down2(indices, batch_size, spatial_shape, ksize, stride, padding, dilation):
    k, s, p, d = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    if k != [2, 2, 2] or s != [2, 2, 2] or p != [0, 0, 0] or d != [1, 1, 1]:
        if k[0] * k[1] * k[2] > 27:
            raise NotImplementedError("doda_amd SparseConv3d supports kernel volumes <= 27, got k=%r" % (k,))
        outids, tbl, tbl_rev, out_shape = _ops.rulebook_conv(indices, spatial_shape, batch_size, k, s, p, d)
        return IndiceData("down2", outids, indices, list(spatial_shape), out_shape, tbl, tbl_rev)
    outids, child, par_off, out_shape = _ops.rulebook_down2(indices, spatial_shape, batch_size)
    return IndiceData("down2", outids, indices, list(spatial_shape), out_shape, child, par_off)


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None, use_hash=False):
    """Upstream-shaped entry: returns (outids, indice_pairs [2,K,N], indice_pair_num [K])."""
    if transpose:
        raise NotImplementedError("transposed sparse convolution is not part of DODA's path")
    if subm:
        data = build_subm(indices, batch_size, spatial_shape, ksize)
    else:
        data = build_down2(indices, batch_size, spatial_shape, ksize, stride, padding, dilation)
    return data.outids, data.indice_pairs, data.indice_pair_num


TILE_MIN_ROWS = 32768    # finest-level rulebooks of at least this many rows get a tilebook (LDS-staged conv kernel)
TILE_KERNEL = os.environ.get("DODA_NO_TILE", "0") != "1"
# Safety valve: the tile kernel lives on the locality of the voxel order (raster-like scans: ~2-3 x 256 distinct
# neighbour rows per 256-row tile).  If more than TILE_OVERFLOW_MAX of a batch's finest-level tiles exceed the
# staging capacity (they are then served from the dense table inside the tile kernel, slower than the dense
# kernel itself), tilebooks are skipped for the next TILE_BACKOFF batches and probed again afterwards.
TILE_OVERFLOW_MAX = 0.25
TILE_BACKOFF = 64
_tile_state = {"skip": 0, "last": None}   # batches still to skip; (tiles, over 64-byte capacity, over list capacity)


def build_pyramid(tensor, n_levels, subm_key="subm%d", down_key="spconv%d", first_level=1, with_pairs=False,
                  with_tiles=None):
    """Build every rulebook of an n-level U-Net up front and store it in `tensor.indice_dict`
    (SubM k3 under subm_key % i, k2s2 under down_key % i), so the convolutions find them cached.
    Rulebooks depend only on the voxel indices; building them before any feature kernel is queued
    means the size read-backs of the strided levels wait on an almost empty stream instead of
    stalling the host in the middle of the forward pass.  with_pairs: also export every rulebook's pair
    lists for the pair-list weight gradient (training with bf16 features)."""
    indices, shape = tensor.indices, tensor.spatial_shape
    # with_tiles: number of finest levels that get a tilebook (True = 2: bf16 rows of 32 / 64 bytes at DODA's
    # widths; 1 for fp32 features); None follows with_pairs (bf16 training)
    n_tile_levels = (2 if with_pairs else 0) if with_tiles is None else (2 if with_tiles is True else int(with_tiles))
    if (_ext is not None and not tensor.indice_dict and indices.is_cuda and indices.dtype == torch.int32
            and indices.shape[0] > 0 and all(int(v) >= 2 for v in shape)):
        tiles_on = TILE_KERNEL and n_tile_levels > 0
        if tiles_on and _tile_state["skip"] > 0:
            _tile_state["skip"] -= 1
            tiles_on = False
        # (one call without the GIL: the builds, their six size read-backs and the tilebook's overflow counters)
        levels, nt, over64, over32 = _ext.build_pyramid_probe(indices, [int(v) for v in shape], int(tensor.batch_size),
                                                              int(n_levels), PAIRS_MIN_ROWS if with_pairs else -1,
                                                              TILE_MIN_ROWS if tiles_on else -1, n_tile_levels)
        if tiles_on and nt >= 0:
            _tile_state["last"] = (nt, over64, over32)
            if over32 > TILE_OVERFLOW_MAX * nt:
                _tile_state["skip"] = TILE_BACKOFF
                # this batch too: plain copies of the tables (no tilebook behind them) keep it on the dense kernels
                levels = [((lv[0].clone() if _ext.has_tilebook(lv[0]) else lv[0]),) + tuple(lv[1:]) for lv in levels]
        for k, (nbr, outids, child, par_off, oshape, sp, sn, sh, dp, dn, dh) in enumerate(levels):   # s*/d*: lists, counts, segments
            lvl = first_level + k
            data = tensor.indice_dict[subm_key % lvl] = IndiceData("subm", indices, indices, list(shape),
                                                                   list(shape), nbr)
            if sp is not None:
                data._wpairs = (sp, sn, sh)
            if k == len(levels) - 1:
                break
            data = tensor.indice_dict[down_key % lvl] = IndiceData("down2", outids, indices, list(shape),
                                                                   list(oshape), child, par_off)
            if dp is not None:
                data._wpairs = (dp, dn, dh)
            indices, shape = outids, list(oshape)
        return tensor.indice_dict
    for lvl in range(first_level, first_level + n_levels):
        key = subm_key % lvl
        if key not in tensor.indice_dict:
            tensor.indice_dict[key] = build_subm(indices, tensor.batch_size, shape, 3)
        if (with_pairs and indices.shape[0] >= PAIRS_MIN_ROWS
                and not (_ext is not None and _ext.has_tilebook(tensor.indice_dict[key].tbl))):   # (tiled: no consumer)
            tensor.indice_dict[key].wgrad_lists()
        if lvl == first_level + n_levels - 1:
            break
        key = down_key % lvl
        data = tensor.indice_dict.get(key)
        if data is None:
            data = build_down2(indices, tensor.batch_size, shape, 2, 2, 0, 1)
            tensor.indice_dict[key] = data
        if with_pairs and WGRAD_PAIRS_DOWN and indices.shape[0] >= PAIRS_MIN_ROWS:
            data.wgrad_lists()
        indices, shape = data.outids, data.out_spatial_shape
    return tensor.indice_dict
