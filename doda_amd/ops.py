"""Tensor-level entry points over the C ABI (include/doda_hip.h).

Each function validates tensors, allocates outputs/workspaces from PyTorch's caching allocator,
and makes exactly the native calls — no arithmetic happens in Python and there is no CPU fallback
(device tensors are required wherever the ABI takes device pointers).
"""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import DodaNativeError, check, lib  # noqa: F401


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter is ~30x cheaper
    than torch.cuda.current_stream(), which dominated the host time of a U-Net step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("doda_amd native ops need device (ROCm) tensors; got a %s tensor — "
                               "there is no CPU fallback" % t.device)


_WS = {}


def _ws(nbytes, device):
    """Scratch for ONE native call.  Calls on a stream execute in order, so a single grow-only
    buffer per (device, stream) is reused instead of allocating per call."""
    nbytes = max(int(nbytes), 256)
    key = (device, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------------------------------
# voxelisation
# ------------------------------------------------------------------------------------------
def voxelize_idx_host(coords, batch_size, mode=4):
    """CPU point->voxel maps; reference PG_OP.voxelize_idx (voxelize.cpp:10-31).

    coords: int64 CPU [N, 3|4].  Returns (output_coords int64 [M,ncol], input_map int32 [N],
    output_map int32 [M, 1+maxActive])."""
    if coords.is_cuda or coords.dtype != torch.int64 or coords.dim() != 2:
        raise RuntimeError("voxelize_idx_host: coords must be a CPU int64 [N,3|4] tensor")
    coords = coords.contiguous()
    n, ncol = coords.shape
    input_map = torch.zeros(n, dtype=torch.int32)
    handle = C.c_void_p()
    n_active, max_active = C.c_int32(), C.c_int32()
    check(lib().doda_voxelize_idx_h(_p(coords), n, ncol, int(batch_size), int(mode), _p(input_map),
                                    C.byref(handle), C.byref(n_active), C.byref(max_active)),
          "doda_voxelize_idx_h")
    out_coords = torch.zeros((n_active.value, ncol), dtype=torch.int64)
    out_map = torch.zeros((n_active.value, max_active.value + 1), dtype=torch.int32)
    check(lib().doda_voxelize_idx_fill_h(handle, _p(coords), _p(out_coords), _p(out_map)),
          "doda_voxelize_idx_fill_h")
    return out_coords, input_map, out_map


def voxelize_idx_device(coords, batch_size, mode=4, sizes=None):
    """Device version of voxelize_idx (same results); coords int64 device [N,3|4].
    sizes = (n_voxels, max_active) known to the caller — the EXACT values a previous call on the same points returned
    (out_map.shape[0], out_map.shape[1] - 1): the D2H read-back of the two output sizes between the assign and the fill
    kernel is skipped.  The outputs are allocated from them; the fill kernels bound every write by them, so a wrong
    size cannot write outside the outputs, but it does not give the reference's result: too small drops the voxels /
    points that do not fit, too large leaves trailing rows empty (count 0, index -1, coordinates 0)."""
    _need_cuda(coords)
    if coords.dtype != torch.int64 or coords.dim() != 2:
        raise RuntimeError("voxelize_idx_device: coords must be int64 [N,3|4]")
    coords = coords.contiguous()
    n, ncol = coords.shape
    dev = coords.device
    input_map = torch.empty(n, dtype=torch.int32, device=dev)      # (every entry is written by the assign kernel)
    counts = torch.empty(2, dtype=torch.int32, device=dev)          # (zeroed by doda_voxelize_idx_assign itself)
    nbytes = lib().doda_voxelize_idx_workspace_bytes(n)
    ws = _ws(nbytes, dev)
    check(lib().doda_voxelize_idx_assign(_p(coords), n, ncol, int(mode), _p(input_map), _p(counts),
                                         _p(ws), ws.numel(), _stream()), "doda_voxelize_idx_assign")
    if sizes is not None:
        m, max_active = int(sizes[0]), int(sizes[1])
    else:
        m, max_active = (int(v) for v in counts.tolist())  # one D2H sync: sizes of the outputs
    max_active = max(max_active, 1)
    # (the fill kernels write every element of both outputs: rows_init all of out_map, finish all of out_coords)
    out_coords = torch.empty((m, ncol), dtype=torch.int64, device=dev)
    out_map = torch.empty((m, max_active + 1), dtype=torch.int32, device=dev)
    if n == 0 or m == 0:
        out_coords.zero_(); out_map.zero_()
    check(lib().doda_voxelize_idx_fill(_p(coords), n, ncol, int(mode), m, max_active,
                                       _p(out_coords), _p(out_map), _p(ws), ws.numel(), _stream()),
          "doda_voxelize_idx_fill")
    return out_coords, input_map, out_map


def _pool_args(feats, out, rules):
    _need_cuda(feats, out, rules)
    if feats.dtype != torch.float32 or out.dtype != torch.float32 or rules.dtype != torch.int32:
        raise RuntimeError("voxel pooling: feats/out must be float32 and rules int32")
    if not (feats.is_contiguous() and out.is_contiguous() and rules.is_contiguous()):
        raise RuntimeError("voxel pooling: tensors must be contiguous")


def voxelize_fp(feats, out, rules, mode, n_active, max_active, n_plane):
    _pool_args(feats, out, rules)
    check(lib().doda_voxelize_fp(_p(feats), _p(out), _p(rules), int(mode), int(n_active),
                                 int(max_active), int(n_plane), _stream()), "doda_voxelize_fp")


def voxelize_fp_rows(feats_a, feats_b, rules, mode, c_out, dtype):
    """The network's input rows in ONE launch (doda_voxelize_fp_rows): voxelization(cat(feats_a, feats_b), rules, mode) cast to
    `dtype` (float32 / bfloat16) and zero-padded to c_out channels.  feats_b may be None.  No gradient."""
    cb = 0 if feats_b is None else feats_b.shape[1]
    for t in (feats_a, feats_b):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 2 or t.device != rules.device):
            raise RuntimeError("voxelize_fp_rows: point features must be contiguous float32 [N, C] on the rules' device")
    if feats_b is not None and feats_b.shape[0] != feats_a.shape[0]:
        raise RuntimeError("voxelize_fp_rows: the two feature matrices differ in rows")
    if rules.dtype != torch.int32 or not rules.is_contiguous() or rules.dim() != 2:
        raise RuntimeError("voxelize_fp_rows: rules must be contiguous int32 [M, 1 + maxActive]")
    if dtype not in (torch.float32, torch.bfloat16) or c_out < feats_a.shape[1] + cb:
        raise RuntimeError("voxelize_fp_rows: dtype float32 / bfloat16, c_out >= the pooled channels")
    m = rules.shape[0]
    out = torch.empty((m, c_out), dtype=dtype, device=rules.device)
    check(lib().doda_voxelize_fp_rows(_p(feats_a), int(feats_a.shape[1]), _p(feats_b) if feats_b is not None else None, int(cb),
                                      _p(rules), int(mode), int(m), int(rules.shape[1] - 1), _p(out), int(c_out),
                                      out.element_size(), _stream()), "doda_voxelize_fp_rows")
    return out


def voxelize_bp(d_out, d_feats, rules, mode, n_active, max_active, n_plane):
    _pool_args(d_out, d_feats, rules)
    check(lib().doda_voxelize_bp(_p(d_out), _p(d_feats), _p(rules), int(mode), int(n_active),
                                 int(max_active), int(n_plane), _stream()), "doda_voxelize_bp")


def point_recover_fp(feats, out, rules, n_active, max_active, n_plane):
    _pool_args(feats, out, rules)
    check(lib().doda_point_recover_fp(_p(feats), _p(out), _p(rules), int(n_active),
                                      int(max_active), int(n_plane), _stream()),
          "doda_point_recover_fp")


def point_recover_bp(d_out, d_feats, rules, n_active, max_active, n_plane):
    _pool_args(d_out, d_feats, rules)
    check(lib().doda_point_recover_bp(_p(d_out), _p(d_feats), _p(rules), int(n_active),
                                      int(max_active), int(n_plane), _stream()),
          "doda_point_recover_bp")


# ------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------
def _shape3(spatial_shape):
    s = [int(v) for v in np.asarray(spatial_shape).reshape(-1)]
    if len(s) != 3:
        raise RuntimeError("spatial_shape must have 3 entries")
    return (C.c_int32 * 3)(*s), s


def _check_indices(indices):
    _need_cuda(indices)
    if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4:
        raise RuntimeError("indices must be int32 [M,4] (batch,x,y,z)")
    return indices.contiguous()


RULEBOOK_GRID_MAX_CELLS = 1 << min(30, max(0, int(os.environ.get("DODA_RULEBOOK_GRID_MAX_LOG2", "28"))))   # csrc/rulebook.hip GRID_MAX_CELLS


def _rulebook_ws(m, batch_size, shape3, device):
    """Workspace of a rulebook build: doda_rulebook_workspace_bytes(m), plus — for grids of up to 2^26 cells — room for the
    direct-address grid behind it (include/doda_hip.h: the build then replaces the hash table by a plain cell -> row array)."""
    nbytes = lib().doda_rulebook_workspace_bytes(m)
    cells = int(batch_size) * int(shape3[0]) * int(shape3[1]) * int(shape3[2])
    if m > 0 and 0 < cells <= RULEBOOK_GRID_MAX_CELLS:
        nbytes = (nbytes + 255) // 256 * 256 + 4 * cells
    return _ws(nbytes, device)


def rulebook_subm(indices, spatial_shape, batch_size, ksize=3):
    """Gather table int32 [ksize^3, M] of a SubMConv3d (nbr[o][t] = input row or -1)."""
    indices = _check_indices(indices)
    m = indices.shape[0]
    shape_c, _ = _shape3(spatial_shape)
    K = ksize ** 3
    nbr = torch.empty((K, m), dtype=torch.int32, device=indices.device)
    ws = _rulebook_ws(m, batch_size, _shape3(spatial_shape)[1], indices.device)
    check(lib().doda_rulebook_subm(_p(indices), m, shape_c, int(batch_size), int(ksize), _p(nbr), m,
                                   _p(ws), ws.numel(), _stream()), "doda_rulebook_subm")
    return nbr


def rulebook_down2(indices, spatial_shape, batch_size):
    """SparseConv3d(k=2,s=2) rulebook.  Returns (out_indices int32 [M_out,4], child int32
    [8,M_out], par_off int32 [8,M], out_shape list[3]).  One D2H sync (M_out)."""
    indices = _check_indices(indices)
    m = indices.shape[0]
    dev = indices.device
    shape_c, s = _shape3(spatial_shape)
    out_shape = [(v - 2) // 2 + 1 for v in s]
    parent = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    off = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
    out_idx_full = torch.empty((max(m, 1), 4), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)   # always written by the scan
    ws = _rulebook_ws(m, batch_size, out_shape, dev)
    check(lib().doda_rulebook_down2_assign(_p(indices), m, shape_c, int(batch_size), _p(parent),
                                           _p(off), _p(out_idx_full), _p(count), _p(ws), ws.numel(),
                                           _stream()), "doda_rulebook_down2_assign")
    m_out = int(count.item())
    child = torch.empty((8, m_out), dtype=torch.int32, device=dev)
    par_off = torch.empty((8, m), dtype=torch.int32, device=dev)
    check(lib().doda_rulebook_down2_tables(_p(parent), _p(off), m, m_out, _p(child), m_out,
                                           _p(par_off), m, _stream()), "doda_rulebook_down2_tables")
    return out_idx_full[:m_out], child, par_off, out_shape   # a view: no copy launch


def _i3(v):
    v = [v] * 3 if isinstance(v, int) else list(v)
    arr = (C.c_int32 * 3)(*[int(x) for x in v])
    return arr


def rulebook_conv(indices, spatial_shape, batch_size, ksize, stride, padding, dilation):
    """Generic SparseConv3d rulebook (K <= 27).  Returns (out_indices int32 [M_out,4], tbl int32
    [K,M_out], tbl_rev int32 [K,M], out_shape list[3]).  One D2H sync (M_out)."""
    indices = _check_indices(indices)
    m = indices.shape[0]
    dev = indices.device
    shape_c, _ = _shape3(spatial_shape)
    k, s, p, d = _i3(ksize), _i3(stride), _i3(padding), _i3(dilation)
    K = k[0] * k[1] * k[2]
    out_shape_c = (C.c_int32 * 3)()
    count = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = lib().doda_rulebook_conv_workspace_bytes(m, K)
    if nbytes == 0:
        raise DodaNativeError("doda_rulebook_conv: kernel volume %d or size not supported" % K)
    ws = _ws(nbytes, dev)
    check(lib().doda_rulebook_conv_assign(_p(indices), m, shape_c, int(batch_size), k, s, p, d, out_shape_c,
                                          _p(count), _p(ws), ws.numel(), _stream()), "doda_rulebook_conv_assign")
    m_out = int(count.item())
    out_idx = torch.empty((m_out, 4), dtype=torch.int32, device=dev)
    tbl = torch.empty((K, m_out), dtype=torch.int32, device=dev)
    tbl_rev = torch.empty((K, m), dtype=torch.int32, device=dev)
    check(lib().doda_rulebook_conv_tables(_p(indices), m, shape_c, int(batch_size), k, s, p, d, m_out,
                                          _p(out_idx), _p(tbl), m_out, _p(tbl_rev), m, _p(ws), ws.numel(),
                                          _stream()), "doda_rulebook_conv_tables")
    return out_idx, tbl, tbl_rev, [int(v) for v in out_shape_c]


def rulebook_subm_generic(indices, spatial_shape, batch_size, ksize):
    """SubM neighbour table for per-axis odd kernel sizes (K <= 27): int32 [K, M]."""
    indices = _check_indices(indices)
    m = indices.shape[0]
    shape_c, _ = _shape3(spatial_shape)
    k = _i3(ksize)
    K = k[0] * k[1] * k[2]
    nbr = torch.empty((K, m), dtype=torch.int32, device=indices.device)
    ws = _ws(lib().doda_rulebook_workspace_bytes(m), indices.device)
    check(lib().doda_rulebook_subm_generic(_p(indices), m, shape_c, int(batch_size), k, _p(nbr), m, _p(ws),
                                           ws.numel(), _stream()), "doda_rulebook_subm_generic")
    return nbr


def rulebook_pairs(tbl, n_rows, flip, pad=True, with_seg=False):
    """spconv-v1.2-format (pairs int32 [2,K,n_rows] -1 padded, pairNum int32 [K]) from a table.
    pad=False leaves the entries past pairNum[o] unwritten (lists for the pair-list weight gradient).
    with_seg: also return the lists' segment prefix int32 [K, ceil(n_rows / 256)] (see doda_hip.h)."""
    _need_cuda(tbl)
    K, ld = tbl.shape
    dev = tbl.device
    pairs = torch.empty((2, K, max(n_rows, 1)), dtype=torch.int32, device=dev)
    pair_num = torch.zeros(K, dtype=torch.int32, device=dev)
    nbytes = lib().doda_rulebook_pairs_workspace_bytes(n_rows, K)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev) if with_seg else _ws(nbytes, dev)
    if n_rows == 0:
        if with_seg:
            return pairs[:, :, :0], pair_num, torch.zeros((K, 1), dtype=torch.int32, device=dev)
        return pairs[:, :, :0], pair_num
    check(lib().doda_rulebook_pairs(_p(tbl), ld, K, n_rows, int(bool(flip)) | (0 if pad else 2), _p(pairs), n_rows,
                                    _p(pair_num), _p(ws), ws.numel(), _stream()),
          "doda_rulebook_pairs")
    if with_seg:
        nt = -(-n_rows // lib().doda_rulebook_pairs_tile())
        return pairs, pair_num, ws[:K * nt * 4].view(torch.int32).view(K, nt)
    return pairs, pair_num


# ------------------------------------------------------------------------------------------
# convolution arithmetic
# ------------------------------------------------------------------------------------------
def _feat_ok(x, name):
    _need_cuda(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("%s must be a contiguous [rows, channels] tensor" % name)


STATS_SLOTS = 8     # DODA_STATS_SLOTS


def totals_zeros(nc, device):
    """Zeroed fp64 totals for nc channels in the library's layout (doda_conv_epilogue.stats_totals): [8 slots, 2 sums, nc / 4
    groups, 16] — a 128-byte line per four channels, the first four doubles of a group used."""
    return torch.zeros((STATS_SLOTS, 2, nc // 4, 16), dtype=torch.float64, device=device)


def totals_sums(totals):
    """[2, nc] float64: the totals summed over their slots."""
    return totals[..., :4].sum(0).reshape(2, -1)


def totals_from_rows(rows):
    """The totals that epilogue rows [n, 2, nc] (fp32) add up to, row k in slot k mod 8 (tests)."""
    n, _, nc = rows.shape
    t = totals_zeros(nc, rows.device)
    for k in range(n):
        t[k % STATS_SLOTS, :, :, :4] += rows[k].double().view(2, nc // 4, 4)
    return t


class _ConvEpilogue(C.Structure):   # doda_conv_epilogue (include/doda_hip.h, ABI 11)
    _fields_ = [("residual", C.c_void_p), ("stats", C.c_void_p), ("stats_rows_h", C.POINTER(C.c_int32)),
                ("bn_x", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_invstd", C.c_void_p), ("bn_gamma", C.c_void_p),
                ("bn_beta", C.c_void_p), ("bn_relu", C.c_int32), ("tilebook_rows", C.c_int32),
                ("tilebook", C.c_void_p), ("residual_bcast", C.c_int32), ("stats_totals", C.c_void_p),
                ("x_ld", C.c_int32), ("y_ld", C.c_int32), ("residual_ld", C.c_int32), ("bn_x_ld", C.c_int32),   # ABI 11: row strides
                ("prologue", C.c_void_p)]                                                                         # ABI 11: doda_conv_prologue *


def tilebook_build(tbl, n_rows=None):
    """Tile-local form of a K = 27 gather table (doda_tilebook_build) as a byte tensor, or None when the
    library has no tilebook for this K.  Pass it to spconv_gather(tilebook=...)."""
    _need_cuda(tbl)
    K, ld = tbl.shape
    n_rows = ld if n_rows is None else int(n_rows)
    nbytes = lib().doda_tilebook_bytes(n_rows, K)
    if nbytes == 0:
        return None
    tb = torch.empty(nbytes, dtype=torch.uint8, device=tbl.device)
    check(lib().doda_tilebook_build(_p(tbl), ld, K, n_rows, _p(tb), nbytes, _stream()), "doda_tilebook_build")
    tb._doda_rows = n_rows
    return tb


class GatherCall:
    """A doda_spconv_gather_ex call prepared once (same arguments as spconv_gather; `out` and, with want_stats, the statistics
    buffer are created once and overwritten by every run) and launched by `run()` with ONE native call: the host cost of the
    extension's call sites.  For timing loops over kernels shorter than the Python wrapper (one 150k-voxel scene: 9 us).
    The call is bound to the stream that was current when the object was made and to the tensors it was made with."""

    def __init__(self, *args, **kwargs):
        self.result = spconv_gather(*args, _call=self, **kwargs)

    def run(self):
        check(lib().doda_spconv_gather_ex(*self._args), "doda_spconv_gather_ex")
        return self.result


def spconv_gather(x, w, tbl, n_out, w_layout, nc, out_f32=False, packed=None, residual=None, tilebook=None,
                  want_stats=False, bn=None, out=None, residual_bcast=False, _call=None):
    """y[t] = sum_o x[tbl[o][t]] @ B_o (doda_spconv_gather_ex, the one gather entry point of ABI 7).  x: [n_in,kc]
    f32|bf16; w: fp32 weights viewed [K,kc,nc] (layout 0) or [K,nc,kc] (layouts 1,2), or `packed`: the
    fragment-packed buffer produced by PackPlan for this (weight, layout, dtype).  Returns y [n_out, nc] in x.dtype (or
    float32 when out_f32); with `residual` ([n_out, nc], dtype of y) returns conv + residual.
    want_stats: returns (y, stats [rows, 2, nc] fp32): the BatchNorm partial sums of the epilogue
    (doda_conv_epilogue.stats) — (sum y, sum y^2), or with bn = (bn_x, mean, invstd, gamma, beta, relu) the
    BatchNorm-backward sums of a data-grad call.  want_stats = "totals" (or a totals tensor to accumulate into): the
    sums as fp64 totals instead (ABI 9, doda_conv_epilogue.stats_totals; totals_zeros / totals_sums), returns (y, totals).
    out: write into this tensor instead of allocating."""
    _feat_ok(x, "x")
    _need_cuda(tbl)
    K, ld = tbl.shape
    kc = x.shape[1]
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("spconv_gather: unsupported feature dtype %s" % x.dtype)
    esz = 4 if x.dtype == torch.float32 else 2
    if packed is not None:
        need = lib().doda_spconv_gather_workspace_bytes(K, kc, nc, esz)
        if packed.numel() * packed.element_size() < need - 255:
            raise RuntimeError("packed weight buffer too small")
        w_ptr, ws_ptr, ws_n = _p(packed), None, 0
        layout = int(w_layout) | 0x100
    else:
        _need_cuda(w)
        w = w.contiguous()
        if w.dtype != torch.float32 or w.numel() != K * kc * nc:
            raise RuntimeError("weight must be float32 with K*kc*nc = %d elements" % (K * kc * nc))
        ws = _ws(lib().doda_spconv_gather_workspace_bytes(K, kc, nc, esz), x.device)
        w_ptr, ws_ptr, ws_n, layout = _p(w), _p(ws), ws.numel(), int(w_layout)
    ydt = torch.float32 if (x.dtype == torch.float32 or out_f32) else torch.bfloat16
    if residual is not None:
        _need_cuda(residual)
        want = (nc,) if residual_bcast else (n_out, nc)
        if residual.dtype != ydt or tuple(residual.shape) != want or not residual.is_contiguous():
            raise RuntimeError("residual must be a contiguous [n_out, nc] tensor (or [nc] with residual_bcast) in the output dtype")
    y = out if out is not None else torch.empty((n_out, nc), dtype=ydt, device=x.device)
    ep = _ConvEpilogue()
    ep.residual = _p(residual) if residual is not None else None
    ep.residual_bcast = int(bool(residual_bcast and residual is not None))
    if tilebook is not None:
        ep.tilebook = _p(tilebook)
        ep.tilebook_rows = int(getattr(tilebook, "_doda_rows", n_out))
    stats, rows = None, C.c_int32(0)
    totals = None
    if want_stats == "totals" or torch.is_tensor(want_stats):   # ABI 9: fp64 totals [8, 2, nc] (a tensor: accumulate into it)
        totals = want_stats if torch.is_tensor(want_stats) else totals_zeros(nc, x.device)
        if totals.dtype != torch.float64 or tuple(totals.shape) != (STATS_SLOTS, 2, nc // 4, 16) or not totals.is_contiguous():
            raise RuntimeError("spconv_gather: totals must be a contiguous float64 [8, 2, nc / 4, 16] tensor (totals_zeros)")
        ep.stats_totals = _p(totals)
    if want_stats is not False and want_stats is not None:
        stats = torch.empty((int(lib().doda_spconv_stats_capacity(n_out)) if totals is None else 1, 2, nc), dtype=torch.float32,
                            device=x.device)
        ep.stats, ep.stats_rows_h = _p(stats), C.pointer(rows)
        if bn is not None:
            bx, mean, invstd, gamma, beta, relu = bn
            ep.bn_x, ep.bn_mean, ep.bn_invstd, ep.bn_gamma, ep.bn_beta = _p(bx), _p(mean), _p(invstd), _p(gamma), _p(beta)
            ep.bn_relu = int(bool(relu))
    call_args = (_p(x), x.shape[0], kc, esz, w_ptr, nc, _p(tbl), ld, K, n_out, _p(y),
                 int(bool(out_f32)), layout, ws_ptr, ws_n, C.byref(ep), _stream())
    check(lib().doda_spconv_gather_ex(*call_args), "doda_spconv_gather_ex")
    if _call is not None:   # (GatherCall: everything the native call references stays alive with the object)
        _call._args = call_args
        _call._keep = (x, w, tbl, y, ep, rows, stats, residual, tilebook, bn, packed, ws if packed is None else None, totals)
    if totals is not None:
        return y, totals
    return (y, stats[:rows.value]) if want_stats else y


def bn_fwd_final(stats, m, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """The reduction half of a training-mode BatchNorm whose statistics rode in a conv epilogue (doda_bn_fwd_final):
    stats [rows, 2, c] fp32 -> (save_mean, save_invstd); running statistics updated in place when given."""
    _need_cuda(stats)
    rows, _, c = stats.shape
    mean = torch.empty(c, dtype=torch.float32, device=stats.device)
    invstd = torch.empty(c, dtype=torch.float32, device=stats.device)
    check(lib().doda_bn_fwd_final(_p(stats), rows, int(m), c, float(eps), float(momentum),
                                  _p(running_mean) if running_mean is not None else None,
                                  _p(running_var) if running_var is not None else None,
                                  _p(num_batches_tracked) if num_batches_tracked is not None else None,
                                  _p(mean), _p(invstd), _stream()), "doda_bn_fwd_final")
    return mean, invstd


class PackPlan:
    """One-launch pre-pack of many weight tensors (doda_spconv_pack_multi).

    entries: list of (w fp32 device tensor viewed [K,*,*], K, kc, nc, layout, elem_bytes).  Output
    buffers and the device descriptor table are created once; run() re-packs everything in a single
    kernel launch (call it whenever the weights changed)."""

    def __init__(self, entries, device):
        import numpy as np
        l = lib()
        n = len(entries)
        dsz = l.doda_spconv_pack_desc_bytes()
        assert dsz == 48
        desc = np.zeros(n, dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("K", "<i4"), ("kc", "<i4"),
                                           ("nc", "<i4"), ("layout", "<i4"), ("esz", "<i4"),
                                           ("n_chunk", "<i4"), ("NB", "<i4"), ("pad", "<i4")]))
        self.outputs = []
        for k, (w, K, kc, nc, layout, esz) in enumerate(entries):
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.numel() == K * kc * nc):
                raise RuntimeError("PackPlan: weights must be contiguous fp32 device tensors")
            out = torch.empty(l.doda_spconv_gather_workspace_bytes(K, kc, nc, esz), dtype=torch.uint8,
                              device=device)
            self.outputs.append(out)   # (the weights are NOT kept alive: the owner re-plans when they die)
            desc[k] = (w.data_ptr(), out.data_ptr(), K, kc, nc, layout, esz, 0, 0, 0)
        blk_end = np.zeros(n, dtype=np.int32)
        total = C.c_int32()
        check(l.doda_spconv_pack_plan_h(desc.ctypes.data, n, blk_end.ctypes.data, C.byref(total)),
              "doda_spconv_pack_plan_h")
        self.n, self.total = n, int(total.value)
        self.desc_dev = torch.from_numpy(desc.view(np.uint8).copy()).to(device)
        self.blk_dev = torch.from_numpy(blk_end).to(device)

    def run(self):
        check(lib().doda_spconv_pack_multi(_p(self.desc_dev), _p(self.blk_dev), self.n, self.total,
                                           _stream()), "doda_spconv_pack_multi")


class _WgradJob(C.Structure):   # doda_wgrad_job (include/doda_hip.h, ABI 2)
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("tbl", C.c_void_p), ("dw", C.c_void_p),
                ("ca", C.c_int32), ("cb", C.c_int32), ("ld", C.c_int32), ("K", C.c_int32),
                ("n_rows", C.c_int32), ("elem_bytes", C.c_int32),
                ("pair_in", C.c_void_p), ("pair_out", C.c_void_p), ("pair_num", C.c_void_p),
                ("pair_ld", C.c_int32), ("n_a", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
                ("pair_seg", C.c_void_p), ("pair_seg_nt", C.c_int32), ("reserved2", C.c_int32),
                ("tilebook", C.c_void_p)]


WGRAD_ACCUMULATE = 1


class WgradPlan:
    """A doda_spconv_wgrad_multi call prepared once (job descriptors, output tensors, workspace, descriptor buffer) and launched
    by `run()` with ONE native call — what the extension's deferred flush costs the host per call.  spconv_wgrad_multi() builds the
    same call in Python every time (~10 us per job): for timing loops over launches shorter than that (one 150k-voxel scene: 60 us
    for eight layers) the wrapper, not the GPU, paces the loop on a busy host.  `outputs` = the dw tensors (overwritten by every run).
    Launches on torch's current stream at run() time; workspace and descriptor buffer belong to the plan (runs must be stream-ordered)."""

    def __init__(self, jobs):
        self.outputs = spconv_wgrad_multi(jobs, _plan=self)

    def run(self):
        check(lib().doda_spconv_wgrad_multi(C.addressof(self._arr), self._n, _p(self._ws), self._ws.numel(), _p(self._desc),
                                            self._desc.numel(), _stream()), "doda_spconv_wgrad_multi")
        return self.outputs


def spconv_wgrad_multi(jobs, _plan=None):
    """Weight gradients of many layers in one native call (doda_spconv_wgrad_multi).
    jobs: list of (a [*,ca], b [n_rows,cb], tbl int32 [K,ld], n_rows[, pairs[, dw[, tilebook]]]); `tilebook` =
    tilebook_build(tbl) (bf16 16 -> 16, K = 27 jobs then take the LDS-staged kernel); `pairs` = None or
    (pair_in int32 [K,ld_p], pair_out int32 [K,ld_p], pair_num int32 [K] | None, seg int32 [K,nt] | None): bf16 jobs with
    16-multiple channel counts then take the pair-list kernel; `dw` = an existing float32 [K,ca,cb] tensor
    to ACCUMULATE into.  Returns the list of dw tensors (float32 [K, ca, cb]), equal to spconv_wgrad per
    job up to the summation order of the partials."""
    if not jobs:
        return []
    arr = (_WgradJob * len(jobs))()
    outs, keep = [], []
    for k, job in enumerate(jobs):
        a, b, tbl, n_rows = job[:4]
        pairs = job[4] if len(job) > 4 else None
        acc_into = job[5] if len(job) > 5 else None
        tilebook = job[6] if len(job) > 6 else None
        _feat_ok(a, "a")
        _feat_ok(b, "b")
        a, b = a.contiguous(), b.contiguous()
        if a.dtype != b.dtype:
            raise RuntimeError("wgrad_multi: a and b must share a dtype")
        if tbl is not None:
            _need_cuda(tbl)
            K, ld = tbl.shape
        else:
            K, ld = pairs[0].shape[0], int(n_rows)
        if acc_into is not None:
            dw = acc_into
            if dw.dtype != torch.float32 or not dw.is_contiguous() or dw.numel() != K * a.shape[1] * b.shape[1]:
                raise RuntimeError("wgrad_multi: the tensor to accumulate into must be contiguous float32 [K,ca,cb]")
        else:
            dw = torch.empty((K, a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
        arr[k] = _WgradJob(_p(a), _p(b), _p(tbl), _p(dw), a.shape[1], b.shape[1], ld, K, int(n_rows),
                           4 if a.dtype == torch.float32 else 2,
                           None, None, None, 0, a.shape[0], WGRAD_ACCUMULATE if acc_into is not None else 0, 0,
                           None, 0, 0, _p(tilebook) if tilebook is not None else None)
        if tilebook is not None:
            keep.append(tilebook)
        if pairs is not None:
            pin, pout, pnum = pairs[:3]
            seg = pairs[3] if len(pairs) > 3 else None
            if pin.dtype != torch.int32 or pout.dtype != torch.int32 or pin.stride(-1) != 1 or pout.stride(-1) != 1 \
                    or pin.shape != pout.shape or pin.dim() != 2 or pin.stride(0) != pout.stride(0):
                raise RuntimeError("wgrad_multi: pair lists must be int32 [K, ld] with a unit inner stride")
            arr[k].pair_in, arr[k].pair_out = _p(pin), _p(pout)
            arr[k].pair_num = _p(pnum)
            arr[k].pair_ld = pin.stride(0) if pin.shape[0] > 1 else pin.shape[1]
            if seg is not None:
                if not seg.is_cuda or seg.dtype != torch.int32 or seg.dim() != 2 or seg.shape[0] != K \
                        or not seg.is_contiguous():
                    raise RuntimeError("wgrad_multi: the segment prefix must be a device int32 [K, nt] tensor")
                arr[k].pair_seg, arr[k].pair_seg_nt = _p(seg), seg.shape[1]
            keep.append((pin, pout, pnum, seg))
        outs.append(dw)
        keep.append((a, b))
    dev = outs[0].device
    l = lib()
    ws = _ws(l.doda_spconv_wgrad_multi_workspace_bytes(C.addressof(arr), len(jobs)), dev)
    desc = torch.empty(l.doda_spconv_wgrad_multi_desc_bytes(len(jobs)), dtype=torch.uint8, device=dev)
    check(l.doda_spconv_wgrad_multi(C.addressof(arr), len(jobs), _p(ws), ws.numel(), _p(desc), desc.numel(),
                                    _stream()), "doda_spconv_wgrad_multi")
    if _plan is not None:   # (WgradPlan: everything the native call references stays alive with the plan)
        _plan._arr, _plan._n, _plan._ws, _plan._desc, _plan._keep = arr, len(jobs), ws, desc, (keep, list(jobs))
    return outs


def spconv_wgrad(a, b, tbl, n_rows):
    """dw[o] = sum_t a[tbl[o][t]]^T b[t]  ->  float32 [K, ca, cb] (one job of doda_spconv_wgrad_multi)."""
    _feat_ok(a, "a")
    _feat_ok(b, "b")
    _need_cuda(tbl)
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("spconv_wgrad: a and b must share a dtype (float32 or bfloat16)")
    if n_rows == 0:
        return torch.zeros((tbl.shape[0], a.shape[1], b.shape[1]), dtype=torch.float32, device=a.device)
    return spconv_wgrad_multi([(a, b, tbl, n_rows)])[0]


def spconv_wgrad_pairs(a, b, pair_in, pair_out, pair_num, pair_seg=None, accumulate_into=None):
    """dw[o] (+)= sum_{p < pair_num[o]} a[pair_in[o,p]]^T b[pair_out[o,p]] -> float32 [K, ca, cb]: one pair-list job
    of doda_spconv_wgrad_multi (bf16 operands, channel counts multiples of 16).  pair_num None: every list is full
    (the identity list of a 1x1 convolution).  pair_seg: the lists' segment prefix int32 [K, nt] (third result of
    rulebook_pairs; required whenever pair_num is given)."""
    _feat_ok(a, "a")
    _feat_ok(b, "b")
    _need_cuda(pair_in, pair_out)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.shape[1] % 16 or b.shape[1] % 16:
        raise RuntimeError("spconv_wgrad_pairs: bf16 operands with channel counts that are multiples of 16")
    return spconv_wgrad_multi([(a, b, None, b.shape[0], (pair_in, pair_out, pair_num, pair_seg), accumulate_into)])[0]


def maxpool_fwd(x, tbl, n_out):
    _feat_ok(x, "x")
    K, ld = tbl.shape
    y = torch.empty((n_out, x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib().doda_maxpool_fwd_f32(_p(x), x.shape[1], _p(tbl), ld, K, n_out, _p(y), _stream()),
          "doda_maxpool_fwd_f32")
    return y


def maxpool_bwd(x, y, dy, tbl, n_out):
    K, ld = tbl.shape
    dx = torch.zeros_like(x)
    check(lib().doda_maxpool_bwd_f32(_p(x), _p(y), _p(dy), x.shape[1], _p(tbl), ld, K, n_out,
                                     _p(dx), _stream()), "doda_maxpool_bwd_f32")
    return dx


# ------------------------------------------------------------------------------------------
# fused BatchNorm(+ReLU)
# ------------------------------------------------------------------------------------------
def _esz(t):
    if t.dtype == torch.float32:
        return 4
    if t.dtype == torch.bfloat16:
        return 2
    raise RuntimeError("unsupported feature dtype %s" % t.dtype)


def bn_relu_fwd(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
                num_batches_tracked=None):
    """-> (y, save_mean, save_invstd).  x: contiguous [m, c] fp32|bf16 on device, c % 4 == 0."""
    _feat_ok(x, "x")
    m, c = x.shape
    y = torch.empty_like(x)
    if training:
        save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    else:
        save_mean = running_mean.float().contiguous()
        save_invstd = torch.rsqrt(running_var.float() + eps).contiguous()
    ws = _ws(lib().doda_bn_workspace_bytes(m, c), x.device)
    rm = _p(running_mean) if (training and running_mean is not None) else None
    rv = _p(running_var) if (training and running_var is not None) else None
    nbt = _p(num_batches_tracked) if (training and num_batches_tracked is not None) else None
    check(lib().doda_bn_relu_fwd(_p(x), m, c, _esz(x), float(eps), float(momentum), _p(gamma), _p(beta),
                                 rm, rv, nbt, int(bool(training)), int(bool(relu)), _p(y), _p(save_mean),
                                 _p(save_invstd), _p(ws), ws.numel(), _stream()), "doda_bn_relu_fwd")
    return y, save_mean, save_invstd


def bn_relu_bwd(x, dy, save_mean, save_invstd, gamma, beta, relu):
    """-> (dx, dgamma, dbeta) for the training-mode forward above."""
    _feat_ok(x, "x")
    _feat_ok(dy, "dy")
    m, c = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _ws(lib().doda_bn_workspace_bytes(m, c), x.device)
    check(lib().doda_bn_relu_bwd(_p(x), _p(dy), m, c, _esz(x), _p(save_mean), _p(save_invstd),
                                 _p(gamma), _p(beta), int(bool(relu)), _p(dx), _p(dgamma), _p(dbeta),
                                 _p(ws), ws.numel(), _stream()), "doda_bn_relu_bwd")
    return dx, dgamma, dbeta


def _add_ld(add, m, c, what):
    """(pointer, row stride in elements) of the second-gradient operand: dense, or a column slice of a wider matrix."""
    _need_cuda(add)
    if add.dim() != 2 or tuple(add.shape) != (m, c) or add.stride(1) != 1 or add.stride(0) < c:
        raise RuntimeError("%s: add must be [m, c] with unit column stride" % what)
    return _p(add), int(add.stride(0))


def bn_relu_bwd_add(x, dy, save_mean, save_invstd, gamma, beta, relu, add):
    """bn_relu_bwd with a second gradient of x summed into dx inside the apply pass (doda_bn_relu_bwd_add); `add`
    may be a column slice of a wider matrix (e.g. g[:, :c] of torch.cat's gradient): no copy is made."""
    _feat_ok(x, "x")
    _feat_ok(dy, "dy")
    m, c = x.shape
    ap, ld = _add_ld(add, m, c, "bn_relu_bwd_add")
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _ws(lib().doda_bn_workspace_bytes(m, c), x.device)
    check(lib().doda_bn_relu_bwd_add(_p(x), _p(dy), m, c, _esz(x), _p(save_mean), _p(save_invstd), _p(gamma), _p(beta),
                                        int(bool(relu)), ap, ld, _p(dx), _p(dgamma), _p(dbeta), _p(ws), ws.numel(),
                                        _stream()), "doda_bn_relu_bwd_add")
    return dx, dgamma, dbeta


def bn_relu_bwd_stats(x, dy, stats, save_mean, save_invstd, gamma, beta, relu, add=None):
    """BatchNorm(+ReLU) backward over the (sum dz, sum dz * xhat) rows of a data-grad conv epilogue
    (doda_bn_relu_bwd_stats); add as in bn_relu_bwd_add.  -> (dx [+ add], dgamma, dbeta)."""
    _feat_ok(x, "x")
    _feat_ok(dy, "dy")
    _need_cuda(stats)
    m, c = x.shape
    ap, ld = _add_ld(add, m, c, "bn_relu_bwd_stats") if add is not None else (None, c)
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    coef = torch.empty(3 * c, dtype=torch.float32, device=x.device)
    check(lib().doda_bn_relu_bwd_stats(_p(x), _p(dy), m, c, _esz(x), _p(stats), stats.shape[0], _p(save_mean),
                                          _p(save_invstd), _p(gamma), _p(beta), int(bool(relu)), ap, ld, _p(dx), _p(dgamma),
                                          _p(dbeta), _p(coef), _stream()), "doda_bn_relu_bwd_stats")
    return dx, dgamma, dbeta


def bn_relu_fwd_totals(x, totals, gamma, beta, running_mean, running_var, momentum, eps, relu, num_batches_tracked=None,
                       totals_b=None):
    """Training-mode BatchNorm(+ReLU) over the fp64 totals of conv epilogues (doda_bn_relu_fwd_totals, ONE launch);
    totals_b: totals of the producer of the trailing columns (a channel concatenation).  -> (y, save_mean, save_invstd)."""
    _feat_ok(x, "x")
    _need_cuda(totals)
    m, c = x.shape
    c_a = int(totals.shape[2]) * 4
    y = torch.empty_like(x)
    save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
    save_invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib().doda_bn_relu_fwd_totals(_p(x), m, c, _esz(x), _p(totals), _p(totals_b) if totals_b is not None else None, c_a,
                                        float(eps), float(momentum), _p(gamma), _p(beta),
                                        _p(running_mean) if running_mean is not None else None,
                                        _p(running_var) if running_var is not None else None,
                                        _p(num_batches_tracked) if num_batches_tracked is not None else None, int(bool(relu)),
                                        _p(y), _p(save_mean), _p(save_invstd), _stream()), "doda_bn_relu_fwd_totals")
    return y, save_mean, save_invstd


def bn_relu_bwd_totals(x, dy, totals, save_mean, save_invstd, gamma, beta, relu, add=None):
    """BatchNorm(+ReLU) backward over the fp64 totals of a data-grad conv epilogue (doda_bn_relu_bwd_totals, ONE launch).
    -> (dx [+ add], dgamma, dbeta)."""
    _feat_ok(x, "x")
    _feat_ok(dy, "dy")
    _need_cuda(totals)
    m, c = x.shape
    ap, ld = _add_ld(add, m, c, "bn_relu_bwd_totals") if add is not None else (None, c)
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib().doda_bn_relu_bwd_totals(_p(x), _p(dy), m, c, _esz(x), _p(totals), _p(save_mean), _p(save_invstd), _p(gamma),
                                        _p(beta), int(bool(relu)), ap, ld, _p(dx), _p(dgamma), _p(dbeta), _stream()),
          "doda_bn_relu_bwd_totals")
    return dx, dgamma, dbeta


def cast_colsum(x):
    """(x.bfloat16(), x.sum(0)) of a device fp32 [n, c <= 64] matrix in one pass (doda_cast_colsum_f32_bf16)."""
    _feat_ok(x, "x")
    n, c = x.shape
    y = torch.empty((n, c), dtype=torch.bfloat16, device=x.device)
    nb = int(lib().doda_cast_colsum_blocks(n, c))
    if x.dtype != torch.float32 or nb == 0:
        raise RuntimeError("cast_colsum: fp32 [n, c <= 64]")
    partial = torch.empty((nb, c), dtype=torch.float32, device=x.device)
    check(lib().doda_cast_colsum_f32_bf16(_p(x), n, c, _p(y), _p(partial), nb, _stream()), "doda_cast_colsum_f32_bf16")
    return y, partial.sum(0)


def pad_channels(x, c_out):
    """x [n, c_in] (fp32 / bf16, contiguous) zero-padded to c_out channels in one kernel (doda_pad_channels)."""
    _feat_ok(x, "x")
    n, c_in = x.shape
    y = torch.empty((n, c_out), dtype=x.dtype, device=x.device)
    check(lib().doda_pad_channels(_p(x), n, c_in, c_out, _esz(x), _p(y), _stream()), "doda_pad_channels")
    return y


# ------------------------------------------------------------------------------------------
# neighbour queries
# ------------------------------------------------------------------------------------------
def knnquery(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    """pointops2_cuda.knnquery_cuda signature (offset/new_offset are END offsets, int32)."""
    _need_cuda(xyz, new_xyz, offset, new_offset, idx, dist2)
    check(lib().doda_knnquery(int(m), int(nsample), _p(xyz), _p(new_xyz), _p(offset),
                              _p(new_offset), int(offset.numel()), _p(idx), _p(dist2), _stream()),
          "doda_knnquery")


def knn_batch(xyz, query_xyz, batch_idxs, query_batch_offsets, idx, n, m, k):
    _need_cuda(xyz, query_xyz, batch_idxs, query_batch_offsets, idx)
    check(lib().doda_knn_batch(int(n), int(m), int(k), _p(xyz), _p(query_xyz), _p(batch_idxs),
                               _p(query_batch_offsets), _p(idx), _stream()), "doda_knn_batch")


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, mean_active, radius):
    _need_cuda(xyz, batch_idxs, batch_offsets, idx, start_len)
    ws = _ws(lib().doda_ballquery_workspace_bytes(int(n)), xyz.device)
    total = C.c_int32(0)
    check(lib().doda_ballquery_batch_p(int(n), int(mean_active), float(radius), _p(xyz),
                                       _p(batch_idxs), _p(batch_offsets), _p(idx), _p(start_len),
                                       C.byref(total), _p(ws), ws.numel(), _stream()),
          "doda_ballquery_batch_p")
    return int(total.value)


# ------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------
def seg_meters(hist, preds, labels, n_classes, ignore_index, p2v=None):
    """hist int64 [3, k] += (intersection, prediction area, target area) of the points (doda_seg_meters; reference
    util/common_utils.py:233-246).  preds: int32 / int64 class per point, or per VOXEL with p2v (int32 [N]) mapping points to voxels."""
    _need_cuda(preds)
    _need_cuda(labels)
    if (hist.dtype != torch.int64 or not hist.is_contiguous() or tuple(hist.shape) != (3, int(n_classes)) or labels.dtype != torch.int64
            or preds.dtype not in (torch.int32, torch.int64) or (p2v is not None and p2v.dtype != torch.int32)):
        raise RuntimeError("seg_meters: hist int64 [3,k], labels int64 [N], preds int32|int64, p2v int32")
    preds, labels = preds.contiguous(), labels.contiguous()
    if p2v is not None:
        _need_cuda(p2v)
        p2v = p2v.contiguous()
    n = labels.shape[0]
    if (p2v.shape[0] if p2v is not None else preds.shape[0]) != n:
        raise RuntimeError("seg_meters: one prediction (or one p2v entry) per label")
    check(lib().doda_seg_meters(_p(preds), 1 if preds.dtype == torch.int64 else 0, _p(p2v) if p2v is not None else None, _p(labels),
                                n, int(n_classes), int(ignore_index), _p(hist), _stream()), "doda_seg_meters")
    return hist


def cross_entropy_fwd(logits, labels, ignore_index):
    """-> (out float32 [2] = {mean loss over valid points, n_valid}, lse float32 [N])."""
    _need_cuda(logits)
    _need_cuda(labels)
    if logits.dtype != torch.float32 or labels.dtype != torch.int64 or logits.dim() != 2:
        raise RuntimeError("cross_entropy: logits float32 [N,C], labels int64 [N]")
    logits, labels = logits.contiguous(), labels.contiguous()
    n, c = logits.shape
    lse = torch.empty(n, dtype=torch.float32, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    ws = _ws(lib().doda_cross_entropy_workspace_bytes(n), logits.device)
    check(lib().doda_cross_entropy_fwd(_p(logits), _p(labels), n, c, int(ignore_index), _p(lse), _p(out), _p(ws),
                                       ws.numel(), _stream()), "doda_cross_entropy_fwd")
    return out, lse


def cross_entropy_bwd(logits, labels, lse, out, grad, ignore_index):
    n, c = logits.shape
    d = torch.empty_like(logits)
    check(lib().doda_cross_entropy_bwd(_p(logits), _p(labels), _p(lse), _p(out), _p(grad), n, c, int(ignore_index),
                                       _p(d), _stream()), "doda_cross_entropy_bwd")
    return d


# ---- point head + loss at voxel level (ABI 11, csrc/head.hip) -----------------------------------------------------
def head_ce_fwd(feats, weight, bias, v2p, labels, ignore_index, want_pred=True):
    """Linear head + cross-entropy over the voxel -> point lists without the [points, classes] matrix (doda_head_ce_fwd).
    -> (out float32 [2] = {mean loss over the valid points, n_valid}, pred int32 [m] = argmax class per voxel or None)."""
    _feat_ok(feats, "feats")
    for t in (weight, v2p, labels):
        _need_cuda(t)
    if weight.dtype != torch.float32 or v2p.dtype != torch.int32 or labels.dtype != torch.int64 or v2p.dim() != 2:
        raise RuntimeError("head_ce: weight float32 [n_cls, c], v2p int32 [m, 1 + max_active], labels int64 [points]")
    feats, weight, v2p, labels = feats.contiguous(), weight.contiguous(), v2p.contiguous(), labels.contiguous()
    m, c = feats.shape
    nb = int(lib().doda_head_ce_blocks(m))
    out = torch.empty(2, dtype=torch.float32, device=feats.device)
    pred = torch.empty(m, dtype=torch.int32, device=feats.device) if want_pred else None
    ws = torch.empty(2 * nb, dtype=torch.float32, device=feats.device)
    check(lib().doda_head_ce_fwd(_p(feats), m, c, feats.element_size(), _p(weight), _p(bias.contiguous()) if bias is not None else None,
                                 weight.shape[0], _p(v2p), v2p.shape[1], _p(labels), int(ignore_index), _p(out),
                                 _p(pred) if pred is not None else None, _p(ws), nb, _stream()), "doda_head_ce_fwd")
    return out, pred


def head_ce_bwd(feats, weight, bias, v2p, labels, ignore_index, out, grad):
    """-> (d_feats [m, c], dz [m, n_cls] — both in the features' dtype —, d_bias float32 [n_cls])."""
    feats, weight, v2p, labels = feats.contiguous(), weight.contiguous(), v2p.contiguous(), labels.contiguous()
    m, c = feats.shape
    n_cls = weight.shape[0]
    nb = int(lib().doda_head_ce_blocks(m))
    d_feats = torch.empty_like(feats)
    dz = torch.empty((m, n_cls), dtype=feats.dtype, device=feats.device)
    dbp = torch.empty((nb, n_cls), dtype=torch.float32, device=feats.device)
    check(lib().doda_head_ce_bwd(_p(feats), m, c, feats.element_size(), _p(weight), _p(bias.contiguous()) if bias is not None else None,
                                 n_cls, _p(v2p), v2p.shape[1], _p(labels), int(ignore_index), _p(out), _p(grad.contiguous()),
                                 _p(d_feats), _p(dz), _p(dbp), nb, _stream()), "doda_head_ce_bwd")
    return d_feats, dz, dbp.sum(0)


def head_dw(feats, dz):
    """d weight [n_cls, c] of the voxel-level head from feats [m, c] and dz [m, n_cls]: the bf16 MFMA kernel (doda_head_dw_bf16) or,
    fp32, the library's weight gradient over an identity table."""
    m, c = feats.shape
    n_cls = dz.shape[1]
    if feats.dtype == torch.bfloat16 and c == 16 and n_cls <= 32:
        nb = int(lib().doda_head_dw_blocks(m))
        part = torch.empty((nb, 32, 16), dtype=torch.float32, device=feats.device)
        check(lib().doda_head_dw_bf16(_p(feats.contiguous()), _p(dz.contiguous()), m, c, n_cls, _p(part), nb, _stream()), "doda_head_dw_bf16")
        return part.sum(0)[:n_cls]
    ident = torch.arange(m, dtype=torch.int32, device=feats.device).view(1, m)
    return spconv_wgrad(feats.contiguous(), dz, ident, m)[0].t()


# ---- optimizer step -------------------------------------------------------------------------------
class _SgdTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("buf", C.c_void_p), ("n", C.c_int64),
                ("first_step", C.c_int32), ("reserved", C.c_int32)]


def sgd_multi(params, grads, bufs, first, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False,
              maximize=False):
    """torch.optim.SGD's update of all (param, grad, momentum buffer) triples in one launch (doda_sgd_multi).
    fp32 contiguous device tensors; bufs[k] may be None when momentum == 0; first[k]: buffer not initialised."""
    n = len(params)
    if n == 0:
        return
    dev = params[0].device
    arr = (_SgdTensor * n)()
    for k, (p, g, b) in enumerate(zip(params, grads, bufs)):
        for t in (p, g) + ((b,) if b is not None else ()):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or not t.is_cuda:
                raise ValueError("sgd_multi needs contiguous fp32 tensors on one HIP device")
        if g.shape != p.shape or (b is not None and b.shape != p.shape):
            raise ValueError("sgd_multi: shape mismatch")
        arr[k].p, arr[k].g, arr[k].buf = p.data_ptr(), g.data_ptr(), (b.data_ptr() if b is not None else None)
        arr[k].n, arr[k].first_step = p.numel(), int(bool(first[k]))
    nbytes = lib().doda_sgd_multi_desc_bytes(n)
    desc = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().doda_sgd_multi(C.cast(arr, C.c_void_p), n, float(lr), float(momentum), float(dampening),
                               float(weight_decay), int(bool(nesterov)), int(bool(maximize)), _p(desc), nbytes,
                               _stream()), "doda_sgd_multi")


# ---- coarse-level op lists (ABI 11, csrc/layers.hip: doda_layers_run) --------------------------------------
CX_GEMM, CX_BNFWD, CX_BNBWD, CX_STATS = 1, 2, 3, 4
CX_F_BARRIER, CX_F_IDENTITY, CX_F_RELU, CX_F_TRAINING, CX_F_ACCUM = 1, 2, 4, 8, 16


class _CxOp(C.Structure):   # doda_cx_op
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("rows", C.c_int32), ("rows_in", C.c_int32),
                ("c_in", C.c_int32), ("c_out", C.c_int32), ("K", C.c_int32), ("tbl_ld", C.c_int32),
                ("x_ld", C.c_int32), ("y_ld", C.c_int32), ("res_ld", C.c_int32), ("aux_ld", C.c_int32),
                ("y2_ld", C.c_int32), ("n_part", C.c_int32), ("c_split", C.c_int32), ("reserved", C.c_int32),
                ("eps", C.c_float), ("momentum", C.c_float),
                ("x", C.c_void_p), ("w", C.c_void_p), ("tbl", C.c_void_p), ("y", C.c_void_p), ("y2", C.c_void_p),
                ("res", C.c_void_p), ("aux", C.c_void_p), ("stats", C.c_void_p), ("stats_b", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("nbt", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("tilebook", C.c_void_p)]


def _cx_array(ops, n_part):
    n = len(ops)
    arr = (_CxOp * n)()
    ptr_fields = {"x", "w", "tbl", "y", "y2", "res", "aux", "stats", "stats_b", "gamma", "beta", "mean", "invstd",
                  "running_mean", "running_var", "nbt", "dgamma", "dbeta"}
    for k, o in enumerate(ops):
        for name, v in o.items():
            if name in ptr_fields:
                if v is not None:
                    _need_cuda(v)
                setattr(arr[k], name, v.data_ptr() if v is not None else None)
            else:
                setattr(arr[k], name, v)
        arr[k].n_part = n_part
    return arr


def stats_totals(c, device):
    """Zeroed fp64 totals for the statistics of a c-channel tensor (the `stats` of an op of layers_run) = totals_zeros."""
    return totals_zeros(c, device)


def layers_run(ops, device, elem_bytes=2):
    """Run a list of ops (dicts keyed by the doda_cx_op field names; tensors for the pointer fields) through the per-layer backend (doda_layers_run, ABI 11): one whole-chip launch
    per op, BatchNorm ops folded into the next convolution's gather where their rows are few (set_pre_rows).  `stats` /
    `stats_b` are fp64 totals (stats_totals).  Returns the number of launches issued."""
    n = len(ops)
    arr = _cx_array(ops, 0)
    launches = C.c_int32(0)
    check(lib().doda_layers_run(C.cast(arr, C.c_void_p), n, int(elem_bytes), C.byref(launches), _stream()), "doda_layers_run")
    return int(launches.value)


def set_pre_rows(fwd=None, bwd=None):
    """Row thresholds of layers_run's BatchNorm folding (doda_set_option DODA_OPT_PRE_FWD_ROWS / _BWD_ROWS; 0: never fold).
    Returns the previous pair."""
    from ._lib import OPT_PRE_BWD_ROWS, OPT_PRE_FWD_ROWS
    old = (int(lib().doda_get_option(OPT_PRE_FWD_ROWS)), int(lib().doda_get_option(OPT_PRE_BWD_ROWS)))
    if fwd is not None:
        check(lib().doda_set_option(OPT_PRE_FWD_ROWS, int(fwd)), "doda_set_option")
    if bwd is not None:
        check(lib().doda_set_option(OPT_PRE_BWD_ROWS, int(bwd)), "doda_set_option")
    return old
