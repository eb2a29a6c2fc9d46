"""Fused BatchNorm1d(+ReLU) for [M, C] sparse-tensor features, backed by libdoda_hip.so.

spconv's SparseSequential applies non-sparse modules to `.features`; DODA's network does that with
`BatchNorm1d(eps=1e-4, momentum=0.1)` followed by `ReLU` in front of every convolution (reference
model/unet.py:28,42-45; model/unet_block.py:23-30,46-49,67-79).  doda_amd.spconv.SparseSequential
recognises that pair (exact torch BatchNorm1d, or DSNorm: reference model/dsnorm.py) and routes
it here; semantics are torch.nn.BatchNorm1d's: batch statistics + running-stat update in training,
running statistics in eval, same gradients.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import ops as _ops
from ._ext import ext as _ext


class _BNReLU(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, training, momentum, eps, relu):
        x = x.contiguous()
        y, mean, invstd = _ops.bn_relu_fwd(x, weight, bias, running_mean, running_var, training,
                                           momentum, eps, relu, nbt)
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.cfg = (bool(training), bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        training, relu = ctx.cfg
        dy = dy.contiguous()
        if training:
            dx, dg, db = _ops.bn_relu_bwd(x, dy, mean, invstd, weight, bias, relu)
        else:  # statistics are constants
            xh = (x.float() - mean) * invstd
            dz = dy.float()
            if relu:
                dz = dz * ((xh * weight + bias) > 0)
            dx = (dz * (weight * invstd)).to(x.dtype)
            dg, db = (dz * xh).sum(0), dz.sum(0)
        return dx, dg.to(weight.dtype), db.to(bias.dtype), None, None, None, None, None, None, None


def _is_dsnorm(m):
    """doda_amd.dsnorm.DSNorm1d or the reference's model/dsnorm.py class (recognised by its buffers)."""
    return (type(m).__name__ in ("DSNorm1d", "DSNorm") and hasattr(m, "running_mean_source")
            and hasattr(m, "running_var_target") and hasattr(m, "domain_label"))


def _running_stats(bn):
    if _is_dsnorm(bn):
        if bn.domain_label:
            return bn.running_mean_target, bn.running_var_target
        return bn.running_mean_source, bn.running_var_source
    return bn.running_mean, bn.running_var


def _module_ok(bn):
    """Structural half of the test, cached on the module (per step this is asked 65 times); the
    parameter dtype is NOT cached — `model.bfloat16()` / `.half()` may change it later —
    and is compared on every call (one attribute read)."""
    ok = bn.__dict__.get("_doda_bn_ok")
    if ok is None:
        ok = bool((type(bn) is nn.BatchNorm1d or _is_dsnorm(bn)) and bn.affine and bn.track_running_stats
                  and bn.momentum is not None)
        bn.__dict__["_doda_bn_ok"] = ok
    return ok and bn._parameters["weight"].dtype == torch.float32


def fusable(bn, features):
    """torch's BatchNorm1d (exact type) or a DSNorm with affine fp32 parameters, running stats and a
    numeric momentum, on a device [M, C] fp32/bf16 matrix with C % 4 == 0."""
    if type(bn) is nn.ReLU or type(bn) is nn.Identity:
        return False
    return (_module_ok(bn) and features.is_cuda and features.dim() == 2
            and features.dtype in (torch.float32, torch.bfloat16) and features.shape[1] % 4 == 0
            and not (bn.training and features.shape[0] < 2))


def batch_norm_relu(features, bn, relu, passthrough=False, stats=None):
    """BatchNorm1d `bn` (+ReLU when `relu`) on features through the fused HIP kernels.
    passthrough: return (y, x_alias) — x_alias is `features` again, for a skip connection; its gradient
    is summed inside this op's backward kernel instead of by an autograd accumulation pass.
    stats: (sum x, sum x^2) partials [rows, 2, C] from the epilogue of the conv that produced `features`
    (doda_spconv_gather_ex): the statistics pass over x is skipped (compiled glue, training mode)."""
    # num_batches_tracked += 1 happens inside the stats kernel (65 one-element add kernels per step
    # otherwise)
    if type(bn) is nn.BatchNorm1d:
        buf = bn._buffers
        running_mean, running_var = buf["running_mean"], buf["running_var"]
    else:
        running_mean, running_var = _running_stats(bn)   # DSNorm: the current domain's pair
    if _ext is not None:
        par = bn._parameters
        stats_b = None
        if isinstance(stats, tuple):   # features = channel concatenation [a | b]: the statistics rows of its two halves
            stats, stats_b = stats
        if passthrough:
            return _ext.bn_relu_pass(features, par["weight"], par["bias"], running_mean, running_var,
                                     bn._buffers["num_batches_tracked"], bn.training, bn.momentum, bn.eps, relu, stats, stats_b)
        return _ext.bn_relu(features, par["weight"], par["bias"], running_mean, running_var,
                            bn._buffers["num_batches_tracked"], bn.training, bn.momentum, bn.eps, relu, stats, stats_b)
    if passthrough:
        return _BNReLU.apply(features, bn.weight, bn.bias, running_mean, running_var,
                             bn.num_batches_tracked, bn.training, bn.momentum, bn.eps, relu), features
    return _BNReLU.apply(features, bn.weight, bn.bias, running_mean, running_var,
                         bn.num_batches_tracked, bn.training, bn.momentum, bn.eps, relu)
