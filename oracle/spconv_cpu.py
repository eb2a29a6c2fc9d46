"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A minimal, independent `spconv`-v1.2-shaped module set on top of oracle/oracle.py (serial C
rulebooks + torch-CPU per-offset gather/mm/scatter), written from spconv's published module
behaviour.  Uses:
  * tests import the REFERENCE's model/unet.py + model/unet_block.py on top of it (in the build
    container, where /root/reference exists) to generate golden logits / gradients and to check
    the layer graph of doda_amd.model;
  * bench.py's cpu_baseline leg times a whole fwd+bwd through it ("port": spconv's CPU path
    cannot be built here, this restatement stands in for it — stated wherever the number appears).
Nothing under doda_amd/ imports this module.
"""
import math
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import oracle as orc


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(v) for v in np.asarray(spatial_shape).reshape(-1)]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)


class SparseModule(nn.Module):
    pass


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def forward(self, input):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                input = m(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = m(input.features)
            else:
                input = m(input)
        return input


class _IndiceConvFn(Function):
    @staticmethod
    def forward(ctx, features, filters, pairs, pair_num, n_out, inverse, subm):
        ctx.save_for_backward(features, filters)
        ctx.rb = (pairs, pair_num, inverse, subm)
        return orc.indice_conv(features, filters, pairs, pair_num, n_out, inverse, subm)

    @staticmethod
    def backward(ctx, grad_out):
        features, filters = ctx.saved_tensors
        pairs, pair_num, inverse, subm = ctx.rb
        d_in, d_w = orc.indice_conv_backward(features, filters, grad_out.contiguous(), pairs,
                                             pair_num, inverse, subm)
        return d_in, d_w, None, None, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 bias=True, subm=False, inverse=False, indice_key=None):
        super().__init__()
        tri = lambda v: [int(x) for x in (v if isinstance(v, (list, tuple)) else [v] * 3)]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = tri(kernel_size), tri(stride)
        self.padding, self.dilation = tri(padding), tri(dilation)
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            nn.init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def forward(self, input):
        feats, idx = input.features, input.indices
        if self.conv1x1:
            out_f = torch.mm(feats, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                out_f = out_f + self.bias
            out = SparseConvTensor(out_f, idx, input.spatial_shape, input.batch_size)
            out.indice_dict = input.indice_dict
            return out
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:  # roles swapped: outputs on the strided conv's saved input sites
            _, outids, pairs, pair_num, out_shape, _ = datas
        elif datas is not None:
            outids, _, pairs, pair_num, _, out_shape = datas
        else:
            np_idx = idx.cpu().numpy().astype(np.int32)
            if self.subm:
                pairs, pair_num = orc.indice_pairs_subm(np_idx, input.batch_size, input.spatial_shape,
                                                        self.kernel_size)
                outids, out_shape = idx, input.spatial_shape
            else:
                oi, pairs, pair_num, out_shape = orc.indice_pairs_conv(
                    np_idx, input.batch_size, input.spatial_shape, self.kernel_size, self.stride,
                    self.padding, self.dilation)
                outids = torch.from_numpy(oi)
            input.indice_dict[self.indice_key] = (outids, idx, pairs, pair_num, input.spatial_shape,
                                                  out_shape)
        out_f = _IndiceConvFn.apply(feats, self.weight, pairs, pair_num, outids.shape[0],
                                    self.inverse, self.subm)
        if self.bias is not None:
            out_f = out_f + self.bias
        out = SparseConvTensor(out_f, outids, out_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, bias,
                         subm=True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, bias,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


def as_module(name="spconv"):
    """A module object exposing this file's classes under spconv's names (for sys.modules)."""
    mod = types.ModuleType(name)
    for k in ("SparseConvTensor", "SparseModule", "SparseSequential", "SubMConv3d", "SparseConv3d",
              "SparseInverseConv3d", "SparseConvolution"):
        setattr(mod, k, globals()[k])
    sub = types.ModuleType(name + ".modules")
    sub.SparseModule = SparseModule
    sub.SparseSequential = SparseSequential
    mod.modules = sub
    return mod, sub


def install(name="spconv"):
    mod, sub = as_module(name)
    sys.modules[name] = mod
    sys.modules[name + ".modules"] = sub
    return mod
