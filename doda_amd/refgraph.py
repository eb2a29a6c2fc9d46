"""The module tree a DODA checkout runs over the drop-in shims — for timing and testing the ZERO-CHANGE route on the GPU box.

The reference's model files (model/unet.py:15-69, model/unet_block.py:9-100) import `spconv` and run unchanged on
doda_amd.spconv (tests/test_host_and_abi.py runs them in the build container), but they cannot travel to
the GPU box.  This file builds a tree with the same CALL PATTERN against the same public surface — nothing of doda_amd.model's
extensions:
  * plain `spconv.SparseSequential` containers called with the tensor only (no `residual=` argument);
  * a block's skip added IN PLACE to the conv branch's output, `output.features += identity.features` (unet_block.py:33-37),
    which bumps the tensor's version and thereby drops the statistics the last conv's epilogue accumulated;
  * fresh `SparseConvTensor` headers for the identity / skip branches (unet_block.py:33,89);
  * `torch.cat` for the level concatenation (unet_block.py:93), `features[input_map.long()]` + `nn.Linear` for the head
    (unet.py:62-64);
  * no residual-block fast path, no one-call coarse levels (those key on doda_amd.model's own classes).
Module names equal the reference's, so `load_state_dict` moves weights between this tree, doda_amd.model.SparseConvNet and a
reference checkpoint.  bench.py times it as the `reference_graph` sub-record; tests/test_gpu_round6.py checks it against
doda_amd.model.SparseConvNet on the same weights.
"""
import functools
from collections import OrderedDict

import torch
from torch import nn

from . import spconv
from .spconv.modules import SparseModule


def _header(t):
    return spconv.SparseConvTensor(t.features, t.indices, t.spatial_shape, t.batch_size)


class RefResidualBlock(SparseModule):
    def __init__(self, cin, cout, norm_fn, indice_key=None):
        super().__init__()
        skip = nn.Identity() if cin == cout else spconv.SubMConv3d(cin, cout, kernel_size=1, bias=False)
        self.i_branch = spconv.SparseSequential(skip)
        self.conv_branch = spconv.SparseSequential(
            norm_fn(cin), nn.ReLU(), spconv.SubMConv3d(cin, cout, kernel_size=3, padding=1, bias=False, indice_key=indice_key),
            norm_fn(cout), nn.ReLU(), spconv.SubMConv3d(cout, cout, kernel_size=3, padding=1, bias=False, indice_key=indice_key))

    def forward(self, input):
        identity = _header(input)
        output = self.conv_branch(input)
        output.features += self.i_branch(identity).features      # in place, as the reference does
        return output


class RefUBlock(nn.Module):
    def __init__(self, n_planes, norm_fn, block_reps, indice_key_id=1):
        super().__init__()
        self.nPlanes = n_planes
        c = n_planes[0]
        subm_key, down_key = "subm%d" % indice_key_id, "spconv%d" % indice_key_id
        self.blocks = spconv.SparseSequential(OrderedDict(
            ("block%d" % r, RefResidualBlock(c, c, norm_fn, indice_key=subm_key)) for r in range(block_reps)))
        if len(n_planes) > 1:
            nxt = n_planes[1]
            self.conv = spconv.SparseSequential(
                norm_fn(c), nn.ReLU(), spconv.SparseConv3d(c, nxt, kernel_size=2, stride=2, bias=False, indice_key=down_key))
            self.u = RefUBlock(n_planes[1:], norm_fn, block_reps, indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(nxt), nn.ReLU(), spconv.SparseInverseConv3d(nxt, c, kernel_size=2, bias=False, indice_key=down_key))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                ("block%d" % r, RefResidualBlock(c * (2 - r) if r < 2 else c, c, norm_fn, indice_key=subm_key))
                for r in range(block_reps)))

    def forward(self, input):
        output = self.blocks(input)
        identity = _header(output)
        if len(self.nPlanes) > 1:
            decoded = self.deconv(self.u(self.conv(output)))
            output.features = torch.cat((identity.features, decoded.features), dim=1)
            output = self.blocks_tail(output)
        return output


class RefSparseConvNet(nn.Module):
    """Same constructor argument and parameter names as doda_amd.model.SparseConvNet (and the reference's model/unet.py)."""

    def __init__(self, cfg):
        super().__init__()
        bb = cfg.MODEL.BACKBONE
        try:
            n_classes = cfg.COMMON_CLASSES.n_classes
        except AttributeError:
            n_classes = cfg.DATA_CONFIG.DATA_CLASS.n_classes
        m = bb.mid_channel
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.input_conv = spconv.SparseSequential(
            spconv.SubMConv3d(bb.in_channel, m, kernel_size=3, padding=1, bias=False, indice_key="subm1"))
        self.unet = RefUBlock([m * i for i in range(1, 8)], norm_fn, bb.block_reps, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(m), nn.ReLU())
        self.linear = nn.Linear(m, n_classes)
        for mod in self.modules():
            if "BatchNorm" in mod.__class__.__name__:
                mod.weight.data.fill_(1.0)
                mod.bias.data.fill_(0.0)

    def forward(self, input, input_map):
        output = self.output_layer(self.unet(self.input_conv(input)))
        point_feats = output.features[input_map.long()]          # voxel -> point (unet.py:62)
        return self.linear(point_feats.to(self.linear.weight.dtype))


def run_reference_route(cfg, model, batch, device, feature_dtype=torch.float32):
    """The reference's forward glue (model/unet.py:72-99): host / device transfers, `pointgroup_ops.voxelization`, an int32 index
    tensor, the network — every rulebook built inside the convolutions' first use, as spconv does."""
    from . import pointgroup_ops
    voxel_coords = batch["voxel_locs"].to(device, non_blocking=True)
    p2v = batch["p2v_map"].to(device, non_blocking=True)
    v2p = batch["v2p_map"].to(device, non_blocking=True)
    feats = batch["feats"].to(device, non_blocking=True)
    if cfg.MODEL.BACKBONE.use_xyz:
        feats = torch.cat((feats, batch["locs_float"].to(device, non_blocking=True)), 1)
    voxel_feats = pointgroup_ops.voxelization(feats, v2p, cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_mode)
    batch_size = batch["offsets"].numel() - 1
    inp = spconv.SparseConvTensor(voxel_feats.to(feature_dtype), voxel_coords.int(), batch["spatial_shape"], batch_size)
    return model(inp, p2v)
