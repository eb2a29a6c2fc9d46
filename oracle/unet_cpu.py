"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

The SparseConv U-Net layer graph (reference model/unet.py:35-46,58-64; model/unet_block.py:10-100)
restated compactly over oracle/spconv_cpu.py, so the CPU baseline of bench.py and smoke() can run a
whole forward+backward where /root/reference is absent.  tests/test_oracle_unet.py checks it
against the golden output of the reference's own model files (same state-dict keys, same logits).
"""
import functools
from collections import OrderedDict

import torch
from torch import nn

from . import spconv_cpu as sp


def _pre(norm, cin, conv):
    return [norm(cin), nn.ReLU(), conv]


class _Res(sp.SparseModule):
    def __init__(self, cin, cout, norm, key):
        super().__init__()
        self.i_branch = sp.SparseSequential(
            nn.Identity() if cin == cout else sp.SubMConv3d(cin, cout, 1, bias=False))
        self.conv_branch = sp.SparseSequential(
            *_pre(norm, cin, sp.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key=key)),
            *_pre(norm, cout, sp.SubMConv3d(cout, cout, 3, padding=1, bias=False, indice_key=key)))

    def forward(self, x):
        idt = sp.SparseConvTensor(x.features, x.indices, x.spatial_shape, x.batch_size)
        out = self.conv_branch(x)
        out.features = out.features + self.i_branch(idt).features
        return out


class _U(nn.Module):
    def __init__(self, planes, norm, reps, level):
        super().__init__()
        self.planes = planes
        c, key = planes[0], "subm%d" % level
        self.blocks = sp.SparseSequential(OrderedDict(
            ("block%d" % i, _Res(c, c, norm, key)) for i in range(reps)))
        if len(planes) > 1:
            n, dkey = planes[1], "spconv%d" % level
            self.conv = sp.SparseSequential(*_pre(norm, c, sp.SparseConv3d(c, n, 2, stride=2, bias=False, indice_key=dkey)))
            self.u = _U(planes[1:], norm, reps, level + 1)
            self.deconv = sp.SparseSequential(*_pre(norm, n, sp.SparseInverseConv3d(n, c, 2, bias=False, indice_key=dkey)))
            self.blocks_tail = sp.SparseSequential(OrderedDict(
                ("block%d" % i, _Res(c * (2 - i), c, norm, key)) for i in range(reps)))

    def forward(self, x):
        out = self.blocks(x)
        if len(self.planes) > 1:
            keep = out.features
            dec = self.deconv(self.u(self.conv(out)))
            out.features = torch.cat((keep, dec.features), 1)
            out = self.blocks_tail(out)
        return out


class OracleUNet(nn.Module):
    def __init__(self, in_channel=3, mid=16, n_classes=20, reps=2):
        super().__init__()
        norm = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.input_conv = sp.SparseSequential(sp.SubMConv3d(in_channel, mid, 3, padding=1, bias=False, indice_key="subm1"))
        self.unet = _U([mid * i for i in range(1, 8)], norm, reps, 1)
        self.output_layer = sp.SparseSequential(norm(mid), nn.ReLU())
        self.linear = nn.Linear(mid, n_classes)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.data.fill_(1.0)
                m.bias.data.fill_(0.0)

    def forward(self, x, p2v):
        out = self.output_layer(self.unet(self.input_conv(x)))
        return self.linear(out.features[p2v.long()])


def forward_backward(net, batch, voxel_feats=None):
    """One CPU fwd+bwd of the U-Net on a collated batch (voxel mean pooling by the oracle)."""
    from . import oracle as orc
    if voxel_feats is None:
        voxel_feats = torch.from_numpy(orc.voxelize_fp(batch["feats"].numpy(), batch["v2p_map"].numpy(), True))
    dtype = next(net.parameters()).dtype
    inp = sp.SparseConvTensor(voxel_feats.to(dtype), batch["voxel_locs"].int(), batch["spatial_shape"],
                              batch["offsets"].numel() - 1)
    scores = net(inp, batch["p2v_map"])
    loss = torch.nn.functional.cross_entropy(scores, batch["labels"], ignore_index=255)
    loss.backward()
    return scores, loss
