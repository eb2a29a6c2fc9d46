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
import torch
from torch.autograd import Function

from .. import ops as _ops


class _IndiceConv(Function):
    @staticmethod
    def forward(ctx, features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout):
        K = fwd_tbl.shape[0]
        cin, cout = weight.shape[-2], weight.shape[-1]
        w = weight.reshape(K, cin, cout)
        ctx.save_for_backward(features, weight)
        ctx.tables = (fwd_tbl, bwd_tbl, n_out, bwd_layout)
        return _ops.spconv_gather(features.contiguous(), w, fwd_tbl, n_out, 0, cout)

    @staticmethod
    def backward(ctx, grad_output):
        features, weight = ctx.saved_tensors
        fwd_tbl, bwd_tbl, n_out, bwd_layout = ctx.tables
        K = fwd_tbl.shape[0]
        cin, cout = weight.shape[-2], weight.shape[-1]
        dy = grad_output.contiguous()  # reference fork patch llijiang/spconv@740a5b7
        w = weight.reshape(K, cin, cout)
        d_feat = d_w = None
        if ctx.needs_input_grad[0]:
            d_feat = _ops.spconv_gather(dy, w, bwd_tbl, features.shape[0], bwd_layout, cin)
        if ctx.needs_input_grad[1]:
            d_w = _ops.spconv_wgrad(features.contiguous(), dy, fwd_tbl, n_out)
            d_w = d_w.reshape(weight.shape).to(weight.dtype)
        return d_feat, d_w, None, None, None, None


def indice_subm_conv(features, weight, data):
    return _IndiceConv.apply(features, weight, data.tbl, data.tbl, data.outids.shape[0], 2)


def indice_conv(features, weight, data):
    return _IndiceConv.apply(features, weight, data.tbl, data.tbl_rev, data.outids.shape[0], 1)


def indice_inverse_conv(features, weight, data):
    # roles swapped: outputs live on the saved (fine) input indices of the strided conv
    return _IndiceConv.apply(features, weight, data.tbl_rev, data.tbl, data.indices.shape[0], 1)


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
