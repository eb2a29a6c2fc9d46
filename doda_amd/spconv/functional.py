"""Autograd functions behind the three convolution modules (spconv v1.2 `spconv.functional`:
indice_subm_conv / indice_conv / indice_inverse_conv) and indice max-pooling.

All three convolutions are one native pattern (include/doda_hip.h):
    forward   y  = gather(x,  W,   fwd_tbl, layout 0)
    data-grad dx = gather(dy, W^T, bwd_tbl, layout 1 or 2)
    wgt-grad  dW = wgrad(a = x, b = dy, fwd_tbl)
with   SubM   : fwd_tbl = bwd_tbl = nbr,          layout 2 (mirrored offsets)
       down2  : fwd_tbl = child,  bwd_tbl = par_off, layout 1
       inverse: fwd_tbl = par_off, bwd_tbl = child,  layout 1
"""
import os

import torch
from torch.autograd import Function

from .. import ops as _ops
from .._ext import ext as _ext

# Backward of a sparse conv = two independent native ops on the same inputs: the data-grad gather
# (on the critical path of the chain rule) and the weight-grad reduction.  With DODA_OVERLAP_BWD=1
# the weight-grad runs on a side HIP stream, forked after dy is ready and joined before backward
# returns.  Measured on MI355X (in-process A/B, B=4 x 150k voxels) it is 4 % SLOWER than issuing both
# on one stream — the two kernels compete for the same CUs and the fork/join costs two events per
# layer — so it is off by default.
_SIDE = {}
_SERIAL = os.environ.get("DODA_OVERLAP_BWD", "0") != "1"


def set_deferred_wgrad(on):
    """Queue the weight gradients of all sparse convolutions during backward and issue them in one
    multi-layer native call when the backward pass completes (doda_spconv_wgrad_multi: ~8 launches
    instead of two per layer).  Needs the compiled extension; parameters must be fp32 leaves whose
    .grad is None when backward starts (zero_grad(set_to_none=True)), otherwise a layer computes its
    gradient on the spot.  The gradients do not pass through AccumulateGrad hooks: use
    doda_amd.dist.GradAllReduce, not DistributedDataParallel.  Returns whether deferral is active."""
    if _ext is None or not _SERIAL:
        return False
    _ext.set_defer_wgrad(bool(on))
    # the extension's nodes then also deposit BatchNorm gamma / beta gradients themselves and keep no autograd edges to their
    # parameters (203 AccumulateGrad nodes less per U-Net step; DODA_DIRECT_GRADS=0: ordinary edges)
    if hasattr(_ext, "set_direct_grads"):
        _ext.set_direct_grads(bool(on) and os.environ.get("DODA_DIRECT_GRADS", "1") != "0")
    return bool(on)


def _gather(x, w, tbl, n_out, layout, nc, packed, residual=None):
    """spconv_gather through the pre-packed weights; falls back to packing inside the call when the
    native fast path refuses them (unaligned or > 2 GB feature matrices use the generic kernel)."""
    if packed is not None:
        try:
            return _ops.spconv_gather(x, w, tbl, n_out, layout, nc, packed=packed, residual=residual)
        except _ops.DodaNativeError:
            pass
    return _ops.spconv_gather(x, w, tbl, n_out, layout, nc, residual=residual)


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


def _backward_pair(need_dx, need_dw, dgrad_fn, wgrad_fn, device):
    """Run dgrad (current stream) and wgrad (side stream) concurrently; returns (dx, dw)."""
    if _SERIAL or not (need_dx and need_dw):
        return (dgrad_fn() if need_dx else None), (wgrad_fn() if need_dw else None)
    main = torch.cuda.current_stream(device)
    side = _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = wgrad_fn()
    dx = dgrad_fn()
    main.wait_stream(side)
    dw.record_stream(main)
    return dx, dw


def _wgrad(features, dy, fwd_tbl, n_out, pairs):
    """Weight gradient of one layer: the pair-list kernel for bf16 operands with 16-multiple channel
    counts when the rulebook's lists are at hand, else the gather-table kernel."""
    if (pairs is not None and features.dtype == torch.bfloat16 and features.shape[1] % 16 == 0
            and dy.shape[1] % 16 == 0):
        try:
            return _ops.spconv_wgrad_pairs(features, dy, *pairs)
        except _ops.DodaNativeError:
            pass
    return _ops.spconv_wgrad(features, dy, fwd_tbl, n_out)


class _IndiceConv(Function):
    @staticmethod
    def forward(ctx, features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, packed, residual=None, pairs=None):
        K = fwd_tbl.shape[0]
        cin, cout = weight.shape[-2], weight.shape[-1]
        w = weight.reshape(K, cin, cout)
        ctx.save_for_backward(features, weight)
        pk_fwd, pk_bwd = packed if packed is not None else (None, None)
        ctx.tables = (fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_bwd, pairs)
        return _gather(features.contiguous(), w, fwd_tbl, n_out, 0, cout, pk_fwd,
                       None if residual is None else residual.contiguous())

    @staticmethod
    def backward(ctx, grad_output):
        features, weight = ctx.saved_tensors
        fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_bwd, pairs = ctx.tables
        K = fwd_tbl.shape[0]
        cin, cout = weight.shape[-2], weight.shape[-1]
        dy = grad_output.contiguous()  # reference fork patch llijiang/spconv@740a5b7
        w = weight.reshape(K, cin, cout)
        d_feat, d_w = _backward_pair(
            ctx.needs_input_grad[0], ctx.needs_input_grad[1],
            lambda: _gather(dy, w, bwd_tbl, features.shape[0], bwd_layout, cin, pk_bwd),
            lambda: _wgrad(features.contiguous(), dy, fwd_tbl, n_out, pairs).reshape(weight.shape).to(weight.dtype),
            dy.device)
        return d_feat, d_w, None, None, None, None, None, (grad_output if ctx.needs_input_grad[7] else None), None


# BatchNorm statistics in the conv epilogues (needs the compiled glue): a conv that
# is asked for them returns (y, stats) with stats = [rows, 2, Cout] partial sums of (y, y^2), which the
# fused BatchNorm that follows turns into mean / invstd without reading y again; in backward the
# data-grad kernel of a conv whose input came straight out of a fused BatchNorm(+ReLU) accumulates that
# BatchNorm's backward sums the same way (linked inside the extension).  DODA_BN_FUSION=0 switches both off.
BN_FUSION = os.environ.get("DODA_BN_FUSION", "1") == "1" and _ext is not None and _SERIAL
if _ext is not None:
    _ext.set_bn_fusion(BN_FUSION)
STATS_MIN_ROWS = int(os.environ.get("DODA_STATS_MIN_ROWS", "4096"))   # below: the one-launch BatchNorm kernels win (csrc/bn.hip BN_SMALL_ROWS)


def set_bn_fusion(on):
    global BN_FUSION
    BN_FUSION = bool(on) and _ext is not None and _SERIAL
    if _ext is not None:
        _ext.set_bn_fusion(BN_FUSION)
    return BN_FUSION


def _conv(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, packed, residual=None, pairs=None,
          want_stats=False):
    """residual: optional [n_out, Cout] tensor in the output dtype; returns conv + residual with the
    add fused into the kernel's store (the residual's gradient is the incoming gradient).
    pairs: (pair_in [K,ld], pair_out [K,ld], pair_num [K] | None) of the rulebook, for the weight gradient.
    want_stats: return (y, stats | None) instead of y (stats: BatchNorm partials of y, see above)."""
    if _ext is not None and _SERIAL:   # compiled autograd glue (no Python per launch)
        pk_fwd, pk_bwd = packed if packed is not None else (None, None)
        fn = _ext.indice_conv_stats if (want_stats and BN_FUSION and n_out > STATS_MIN_ROWS) else _ext.indice_conv
        if pairs is None:
            out = fn(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_fwd, pk_bwd, residual)
        else:
            out = fn(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_fwd, pk_bwd, residual,
                     pairs[0], pairs[1], pairs[2], pairs[3] if len(pairs) > 3 else None)
        if want_stats:
            return out if fn is _ext.indice_conv_stats else (out, None)
        return out
    y = _IndiceConv.apply(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, packed, residual, pairs)
    return (y, None) if want_stats else y


# The pair-list weight gradient (a pair-list job of doda_spconv_wgrad_multi) needs the rulebook's pair lists (exported
# once per rulebook on the rulebook stream); DODA_WGRAD_PAIRS=0 keeps every layer on the gather-table kernel.
WGRAD_PAIRS = os.environ.get("DODA_WGRAD_PAIRS", "1") == "1"


# (round 5) lists of the STRIDED rulebooks (k2 s2 conv / inverse conv): off by default — once the SubM layers of levels 1-2 took
# the tile weight gradient, these six layers were the only consumers left and the export of their lists (pairs_count / _scan /
# _fill per rulebook: 230 us per step on the rulebook stream) cost four times what the pair-list kernels saved (54 us)
WGRAD_PAIRS_DOWN = os.environ.get("DODA_WGRAD_PAIRS_DOWN", "0") == "1"
PAIRS_MIN_ROWS = 65536   # smaller rulebooks stay on the gather-table kernel: the list export (three launches
#                          per rulebook, issued by a host that is as busy as the GPU) would cost more than it saves


def _want_pairs(features, weight, n_rows=None):
    """bf16 operands, 16-multiple channel counts, gradient recording on, a weight that wants one, and a
    rulebook large enough for the lists to pay off."""
    return (WGRAD_PAIRS and features.dtype == torch.bfloat16 and weight.requires_grad and torch.is_grad_enabled()
            and weight.shape[-2] % 16 == 0 and weight.shape[-1] % 16 == 0
            and (features.shape[0] if n_rows is None else n_rows) >= PAIRS_MIN_ROWS)


def _lists(data, features, weight, inverse=False, n_rows=None):
    """The rulebook's pair lists for a layer's weight gradient, or None: lists that already exist are used; a SubM rulebook
    with a tilebook needs none (its layers take the tile weight gradient or the gather table) and a strided rulebook gets none
    unless DODA_WGRAD_PAIRS_DOWN=1 — nothing is exported just in case (round 5: four such exports per step ran on the
    step's own stream, 230 us of kernels and twelve launches, for jobs that did not read them)."""
    if not _want_pairs(features, weight, n_rows):
        return None
    if data._wpairs is None:
        if data.kind == "subm":
            if _ext is not None and _ext.has_tilebook(data.tbl):
                return None
        elif not WGRAD_PAIRS_DOWN:
            return None
    return data.wgrad_lists(inverse=inverse)


def conv1x1(features, weight, ident, packed=None, want_stats=False):
    """SubMConv3d(kernel_size=1) (upstream: features @ W.view(Cin,Cout)) as a K = 1 gather-GEMM over an
    identity table: the library GEMM picked for these skinny shapes ([600k,32] @ [32,16]) runs 5-10x
    slower than the gather kernel on MI355X.  The identity table doubles as both pair lists."""
    pairs = (ident, ident, None) if _want_pairs(features, weight) else None
    return _conv(features, weight, ident, ident, features.shape[0], 1, packed, None, pairs, want_stats)


def indice_subm_conv(features, weight, data, packed=None, residual=None, want_stats=False):
    pairs = _lists(data, features, weight)
    return _conv(features, weight, data.tbl, data.tbl, data.outids.shape[0], 2, packed, residual, pairs, want_stats)


def indice_conv(features, weight, data, packed=None, want_stats=False):
    pairs = _lists(data, features, weight)
    return _conv(features, weight, data.tbl, data.tbl_rev, data.outids.shape[0], 1, packed, None, pairs, want_stats)


def indice_inverse_conv(features, weight, data, packed=None, want_stats=False):
    # roles swapped: outputs live on the saved (fine) input indices of the strided conv
    pairs = _lists(data, features, weight, inverse=True, n_rows=data.indices.shape[0])
    return _conv(features, weight, data.tbl_rev, data.tbl, data.indices.shape[0], 1, packed, None, pairs, want_stats)


class _IndiceMaxPool(Function):
    @staticmethod
    def forward(ctx, features, tbl, n_out):
        out = _ops.maxpool_fwd(features.contiguous(), tbl, n_out)
        ctx.save_for_backward(features, out)
        ctx.tables = (tbl, n_out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, out = ctx.saved_tensors
        tbl, n_out = ctx.tables
        return _ops.maxpool_bwd(features.contiguous(), out, grad_output.contiguous(), tbl, n_out), None, None


def indice_maxpool(features, data):
    return _IndiceMaxPool.apply(features, data.tbl, data.outids.shape[0])
