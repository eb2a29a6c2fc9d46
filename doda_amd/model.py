"""SparseConv U-Net of DODA, as a caller of doda_amd.spconv.

The reference's own model files (model/unet.py:15-69, model/unet_block.py:9-100) run unchanged on
top of doda_amd.spconv; they cannot travel to the GPU box, so bench.py and
smoke() drive this counterpart.  It has the same layer graph (7 levels, channels
16*i, block_reps residual blocks per stage, BN(eps=1e-4, momentum=0.1) -> ReLU -> conv
pre-activation order, k2s2 down / inverse up sharing `spconv{i}`, concatenation skip, BN weight 1 /
bias 0 init) and the same module names, so state dicts are interchangeable with the reference's
checkpoints (`input_conv.0.weight`, `unet.blocks.block0.conv_branch.2.weight`, `unet.conv.2.weight`,
`unet.u...`, `unet.deconv.2.weight`, `unet.blocks_tail.block0.i_branch.0.weight`, `linear.weight`).
tests/test_oracle_unet.py and tests/test_gpu_unet.py check that equivalence through
tests/golden/unet_golden.npz / unet_state_keys.json, which the reference's own model files produced.
"""
import functools
from collections import OrderedDict
from types import SimpleNamespace

import torch
from torch import nn
from torch.autograd import Function

from torch.nn.modules import module as _torch_module

from . import ops as _ops
from . import pointgroup_ops, spconv
from .spconv import conv as _cv
from .spconv import functional as Fsp
from .spconv.modules import SparseModule


def default_cfg(n_classes=20, in_channel=3, mid_channel=16, block_reps=2, block_residual=True,
                use_xyz=False, voxel_mode=4):
    """The keys SparseConvNet reads, with cfgs/scannet/spconv.yaml's values (:16-21)."""
    backbone = SimpleNamespace(in_channel=in_channel, mid_channel=mid_channel, block_reps=block_reps,
                               block_residual=block_residual, use_xyz=use_xyz)
    return SimpleNamespace(MODEL=SimpleNamespace(BACKBONE=backbone),
                           COMMON_CLASSES=SimpleNamespace(n_classes=n_classes),
                           DATA_CONFIG=SimpleNamespace(
                               DATA_PROCESSOR=SimpleNamespace(voxel_mode=voxel_mode),
                               DATA_CLASS=SimpleNamespace(n_classes=n_classes, ignore_label=255)))


def _shallow(t):
    """A second SparseConvTensor header over the same features / rulebooks: SparseSequential
    re-binds `.features` on the object it is given, so a branch that must keep the incoming
    features takes its own header (reference unet_block.py:33,89)."""
    c = spconv.SparseConvTensor(t.features, t.indices, t.spatial_shape, t.batch_size)
    c.indice_dict = t.indice_dict   # same forward pass: rulebooks and weight-pack generation are shared
    return c


# One extension call per residual block instead of one per BatchNorm / convolution (DODA_FAST_BLOCKS=0: module by module)
import os as _os
FAST_BLOCKS = _os.environ.get("DODA_FAST_BLOCKS", "1") == "1"
# Skip connections through the BatchNorm's pass-through output (round 3): a tensor that feeds a fused BatchNorm AND a skip
# (the U-Net level's concatenation, reference model/unet_block.py:89-93; the 1x1 skip of a channel-changing block) hands
# the skip the BatchNorm op's alias of it, so the skip's gradient — for the concatenation a column slice of the wider
# gradient, taken with its row stride, no copy — is summed inside the BatchNorm's backward kernel: 12 accumulation
# kernels per step less.  DODA_SKIP_FUSION=0: off.
SKIP_IN_BLOCK = SKIP_VIA_BN = _os.environ.get("DODA_SKIP_FUSION", "1") == "1"
# The concatenation's BatchNorm takes the statistics rows of its two halves (conv epilogues) instead of sweeping the new tensor
CAT_STATS = _os.environ.get("DODA_CAT_STATS", "1") == "1"


def _subm3(cin, cout, key):
    return spconv.SubMConv3d(cin, cout, kernel_size=3, padding=1, bias=False, indice_key=key)


class ResidualBlock(SparseModule):
    """y = conv_branch(x) + i_branch(x); i_branch is Identity or a 1x1 SubM when Cin != Cout."""

    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        skip = nn.Identity() if in_channels == out_channels else \
            spconv.SubMConv3d(in_channels, out_channels, kernel_size=1, bias=False)
        self.i_branch = spconv.SparseSequential(skip)
        self.conv_branch = spconv.SparseSequential(
            norm_fn(in_channels), nn.ReLU(), _subm3(in_channels, out_channels, indice_key),
            norm_fn(out_channels), nn.ReLU(), _subm3(out_channels, out_channels, indice_key))

    def forward(self, input):
        # reference: output = conv_branch(input); output.features += i_branch(identity).features
        # (model/unet_block.py:33-37).  Here the add rides in the last convolution's store and, for an
        # identity skip, the skip's gradient in the first BatchNorm's backward kernel.
        identity = type(self.i_branch[0]) is nn.Identity and len(self.i_branch) == 1
        if FAST_BLOCKS:
            out = self._forward_one_call(input, identity)
            if out is not None:
                return out
        if identity:
            return self.conv_branch(input, residual="input")
        skip = self.i_branch(_shallow(input))
        return self.conv_branch(input, residual=skip.features)

    def _plan(self):
        """The block's static operands for ext.residual_block — (generation, bn1 list, bn2 list, bn1, bn2, conv1, conv2, the six modules) —
        or False when the block is not the plain [BN, ReLU, SubM3, BN, ReLU, SubM3] of the reference."""
        plan = self.__dict__.get("_doda_plan")
        if plan is not None and (plan is False or plan[0] == _cv._GEN[0]):
            return plan
        mods = list(self.conv_branch._modules.values())
        ok = (len(mods) == 6 and type(mods[0]) is nn.BatchNorm1d and type(mods[1]) is nn.ReLU
              and type(mods[3]) is nn.BatchNorm1d and type(mods[4]) is nn.ReLU
              and all(type(m) is spconv.SubMConv3d and m.bias is None and m.kernel_size == [3, 3, 3] and not m.inverse
                      and m.in_channels % 4 == 0 and m.out_channels % 4 == 0 for m in (mods[2], mods[5]))
              and mods[2].indice_key is not None and mods[2].indice_key == mods[5].indice_key
              and all(m.affine and m.track_running_stats and m.momentum is not None and not m._forward_hooks
                      and not m._forward_pre_hooks for m in (mods[0], mods[3]))
              and not any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods))
        if not ok:
            plan = False
        else:
            def bn_list(bn):
                return [bn._parameters["weight"], bn._parameters["bias"], bn._buffers["running_mean"],
                        bn._buffers["running_var"], bn._buffers["num_batches_tracked"]]
            plan = (_cv._GEN[0], bn_list(mods[0]), bn_list(mods[3]), mods[0], mods[3], mods[2], mods[5], tuple(mods))
        self.__dict__["_doda_plan"] = plan
        return plan

    def _forward_one_call(self, input, identity):
        """The whole block as ONE extension call (csrc_ext residual_block: the same native ops and autograd nodes the
        module-by-module path issues, without the interpreter between them), or None when any precondition of that
        path does not hold — then the modules run one by one as before."""
        ext = Fsp._ext
        if ext is None or not Fsp._SERIAL or not hasattr(ext, "residual_block"):
            return None
        plan = self._plan()
        if plan is False:
            return None
        _, l1, l2, bn1, bn2, c1, c2, all_mods = plan
        # hooks registered since the plan was made (feature taps, profilers) must fire: module by module then
        _mod = _torch_module   # (imported at module level: the function-level imports were ~5 us per block and step)
        if (_mod._global_forward_hooks or _mod._global_forward_pre_hooks or _mod._global_backward_hooks
                or _mod._global_backward_pre_hooks or self.conv_branch._forward_hooks or self.conv_branch._forward_pre_hooks):
            return None
        for m in all_mods:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return None
        feats = input.features
        training = bn1.training
        if (not feats.is_cuda or feats.dim() != 2 or feats.shape[0] < 2 or feats.dtype not in (torch.float32, torch.bfloat16)
                or bn2.training != training or c1.training != training or c2.training != training
                or l1[0] is not bn1._parameters["weight"] or l2[0] is not bn2._parameters["weight"]
                or l1[2] is not bn1._buffers["running_mean"] or l2[2] is not bn2._buffers["running_mean"]
                or l1[0].dtype != torch.float32 or l2[0].dtype != torch.float32):
            return None
        data = input.find_indice_pair(c1.indice_key)
        if data is None or data.kind != "subm":
            return None
        pk1 = c1._packed(feats, input.indice_dict)
        pk2 = c2._packed(feats, input.indice_dict)
        if pk1 is None or pk2 is None:
            return None
        skip, sc = None, []
        if not identity:
            # the 1x1 skip convolution (reference model/unet_block.py:18-21) runs inside the extension call, on the first
            # BatchNorm's pass-through alias of the input: its data gradient is then summed inside that BatchNorm's
            # backward kernel instead of by an autograd accumulation kernel (SKIP_IN_BLOCK; the input has two consumers)
            sk = self.i_branch[0] if len(self.i_branch) == 1 else None
            if (SKIP_IN_BLOCK and training and type(sk) is spconv.SubMConv3d and sk.conv1x1 and sk.bias is None
                    and sk.training and sk.in_channels % 4 == 0 and sk.out_channels % 4 == 0
                    and sk._parameters["weight"].dtype == torch.float32 and not (sk._forward_hooks or sk._forward_pre_hooks
                                                                              or sk._backward_hooks or sk._backward_pre_hooks)
                    and not (self.i_branch._forward_hooks or self.i_branch._forward_pre_hooks)):
                pks = sk._packed(feats, input.indice_dict)
                if pks is not None:
                    from .spconv.conv import _identity_table
                    ident = _identity_table(feats.shape[0], feats.device)
                    wsk = sk._parameters["weight"]
                    sc = [wsk, pks[0], pks[1], ident, ident if Fsp._want_pairs(feats, wsk) else None]
            if not sc:
                skip = self.i_branch(_shallow(input)).features
                if skip.dtype != feats.dtype:
                    return None
        n_out = data.outids.shape[0]
        w1, w2 = c1._parameters["weight"], c2._parameters["weight"]
        p1, p2 = Fsp._lists(data, feats, w1), Fsp._lists(data, feats, w2)
        if p1 is not None and p2 is not None:
            p = p1
            rb = [data.tbl, p[0], p[1], p[2], p[3] if len(p) > 3 else None]
        elif p1 is not None or p2 is not None:
            return None
        else:
            rb = [data.tbl, None, None, None, None]
        st = input.__dict__.get("_doda_stats")
        stats_in = st[1] if (st is not None and st[0] is feats and st[2] == feats._version) else None
        stats_in_b = None
        if isinstance(stats_in, tuple):   # the input is a channel concatenation: statistics rows of its two halves
            stats_in, stats_in_b = stats_in
        want_stats = training and Fsp.BN_FUSION and n_out > Fsp.STATS_MIN_ROWS
        y, stats = ext.residual_block(feats, stats_in, l1, l2, training, bn1.momentum, bn1.eps, bn2.momentum, bn2.eps,
                                      [w1, pk1[0], pk1[1]], [w2, pk2[0], pk2[1]], rb, n_out, skip, want_stats, sc, stats_in_b)
        out = spconv.SparseConvTensor(y, data.outids, data.out_spatial_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        if stats is not None:
            out._doda_stats = (y, stats, y._version)
        return out


class VGGBlock(SparseModule):
    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        self.conv_layers = spconv.SparseSequential(
            norm_fn(in_channels), nn.ReLU(), _subm3(in_channels, out_channels, indice_key))

    def forward(self, input):
        return self.conv_layers(input)


# Coarse levels as ONE extension call per direction (csrc_ext coarse_ublock: a UBlock subtree compiled into an op list, one autograd
# node, gradients of every layer incl. the deferred weight-gradient jobs), issued by the library as whole-chip per-layer launches
# (doda_layers_run, csrc/layers.hip) with every BatchNorm whose rows are few folded into the gather of the convolution behind it
# and the level concatenations written in place by their two producers: ~60 launches, ~100 autograd nodes and the interpreter
# between them less per step; bf16 and fp32.  Bench step 5.05 -> 4.81 ms, host floor 5.0 -> 4.1 ms (tools/layers_ab.py).
# DODA_COARSE_MODE: "layers" (default) | "off" (module by module).  Round 5's persistent single-XCD executor ("exec") is gone:
# with the per-layer launches issued from inside the library it lost at every batch size (4.30 against 4.05-4.28 ms at the host
# floor, 7.35 against 4.81 at the bench size).
# ABI 12: the op list takes ANY level (a GEMM op carries its table's tilebook, the LDS-staged kernels write column slices), so the
# default subtree is the whole U-Net: one extension call forward, one autograd node backward, no interpreter between the 99 + 134
# launches of a step's convolutions and BatchNorms, the three remaining torch.cat and their gradient slices gone (4.61 -> 4.53 ms,
# host floor 4.1 -> 2.1 ms per step).
COARSE_MODE = _os.environ.get("DODA_COARSE_MODE", "layers")
COARSE_EXEC_LEVEL = int(_os.environ.get("DODA_COARSE_LEVEL", "1"))
COARSE_LAYERS_MAX_ROWS = int(_os.environ.get("DODA_COARSE_LAYERS_MAX_ROWS", str(1 << 23)))


def set_coarse_mode(mode, level=None):
    """mode: "layers" | "off"; level: the U-Net level whose UBlock subtree becomes one extension call."""
    global COARSE_MODE, COARSE_EXEC_LEVEL
    assert mode in ("layers", "off"), mode
    COARSE_MODE = mode
    if level is not None:
        COARSE_EXEC_LEVEL = int(level)
    return mode


def choose_coarse_backend(rows, dtype):
    """"layers" when a subtree whose input has `rows` rows runs as one extension call, None for module by module."""
    if COARSE_MODE == "off" or dtype not in (torch.bfloat16, torch.float32):
        return None
    return "layers" if 2 <= rows <= COARSE_LAYERS_MAX_ROWS else None


def _coarse_entry_level():
    """The level whose UBlock becomes one call: COARSE_EXEC_LEVEL, but not above the levels whose inputs carry tensor hooks (the early
    gradient exchange at EARLY_LEVEL, the side-stream weight gradients at WGRAD_SIDE_LEVEL: both opt-in)."""
    lvl = COARSE_EXEC_LEVEL
    if _early_exchange[0] is not None:
        lvl = max(lvl, EARLY_LEVEL + 1)
    if WGRAD_SIDE_LEVEL:
        lvl = max(lvl, WGRAD_SIDE_LEVEL + 1)
    return lvl


def _bn_list(bn):
    return [bn._parameters["weight"], bn._parameters["bias"], bn._buffers["running_mean"], bn._buffers["running_var"],
            bn._buffers["num_batches_tracked"]]


def _plain_bn(m):
    return (type(m) is nn.BatchNorm1d and m.affine and m.track_running_stats and m.momentum is not None
            and not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks)
            and m._parameters["weight"].dtype == torch.float32)


def _plain_conv(m, cls, ksize):
    return (type(m) is cls and m.bias is None and m.kernel_size == ksize and m._parameters["weight"].dtype == torch.float32
            and m.in_channels % 16 == 0 and m.out_channels % 16 == 0 and m.in_channels >= 16
            and not (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks))


# Early gradient exchange (doda_amd.dist.GradAllReduce.arm): a callback fired from a tensor hook when the backward pass has
# produced the gradient of level EARLY_LEVEL's input — every deeper layer and the whole decoder are done by then.
EARLY_LEVEL = 3
_early_exchange = [None]


def set_early_exchange(fn):
    _early_exchange[0] = fn


# Level at which the forward pass turns launch-floor bound (a few thousand rows and fewer): PyramidPrefetcher starts the
# NEXT batch's rulebook kernels there instead of next to the level-1 convolutions (see PyramidPrefetcher.submit).
COARSE_LEVEL = 4
# Weight gradients under the coarse levels (round 5): when the backward pass produces the gradient of level WGRAD_SIDE_LEVEL's
# OUTPUT — the decoder of every finer level is done, the coarse levels' chain of small kernels starts — the weight gradients
# queued so far are issued on a second stream (extension: flush_wgrads_side) and run under that chain; the flush at the end of
# backward joins them.  Measured (tools/side_ab.sh, side_ab2.sh: 4 and 8 scenes, levels 2-5): no change of the step in either
# regime — the second queue's workgroups do not run under the chain, they delay it —, so 0 = off is the default.
WGRAD_SIDE_LEVEL = int(_os.environ.get("DODA_WGRAD_SIDE_LEVEL", "0"))


def _side_flush_hook(g):
    if Fsp._ext is not None:
        Fsp._ext.flush_wgrads_side()
    return g
_coarse_hooks = []


class UBlock(nn.Module):
    """One U-Net level: blocks -> [down -> UBlock(level+1) -> up -> cat -> blocks_tail]."""

    def __init__(self, nPlanes, norm_fn, block_reps, block, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        self.level = indice_key_id
        c = nPlanes[0]
        subm_key, down_key = "subm%d" % indice_key_id, "spconv%d" % indice_key_id
        self.blocks = spconv.SparseSequential(OrderedDict(
            ("block%d" % r, block(c, c, norm_fn, indice_key=subm_key)) for r in range(block_reps)))
        if len(nPlanes) > 1:
            c_next = nPlanes[1]
            self.conv = spconv.SparseSequential(
                norm_fn(c), nn.ReLU(),
                spconv.SparseConv3d(c, c_next, kernel_size=2, stride=2, bias=False, indice_key=down_key))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, indice_key_id=indice_key_id + 1)
            self.deconv = spconv.SparseSequential(
                norm_fn(c_next), nn.ReLU(),
                spconv.SparseInverseConv3d(c_next, c, kernel_size=2, bias=False, indice_key=down_key))
            self.blocks_tail = spconv.SparseSequential(OrderedDict(
                ("block%d" % r, block(c * (2 - r) if r < 2 else c, c, norm_fn, indice_key=subm_key))
                for r in range(block_reps)))

    # ---- the subtree as executor steps (csrc_ext coarse_ublock) ----
    def _coarse_modules(self):
        """Static part: [("rb", block) | ("down", seq, level) | ("up", seq, level)] in execution order, or False when a
        module of the subtree is not the plain reference form (model/unet_block.py:9-37,55-100)."""
        plan = self.__dict__.get("_doda_coarse")
        if plan is not None:
            return plan
        steps, ok = [], [True]

        def walk(ub):
            for blk in ub.blocks._modules.values():
                steps.append(("rb", blk, ub.level))
            if len(ub.nPlanes) > 1:
                steps.append(("down", ub.conv, ub.level))
                walk(ub.u)
                steps.append(("up", ub.deconv, ub.level))
                for blk in ub.blocks_tail._modules.values():
                    steps.append(("rb", blk, ub.level))

        walk(self)
        for kind, m, _ in steps:
            if kind == "rb":
                mods = list(m.conv_branch._modules.values()) if type(m) is ResidualBlock else []
                sk = list(m.i_branch._modules.values()) if type(m) is ResidualBlock else []
                good = (len(mods) == 6 and _plain_bn(mods[0]) and type(mods[1]) is nn.ReLU and _plain_conv(mods[2], spconv.SubMConv3d, [3, 3, 3])
                        and _plain_bn(mods[3]) and type(mods[4]) is nn.ReLU and _plain_conv(mods[5], spconv.SubMConv3d, [3, 3, 3])
                        and mods[2].indice_key == mods[5].indice_key and len(sk) == 1
                        and (type(sk[0]) is nn.Identity or _plain_conv(sk[0], spconv.SubMConv3d, [1, 1, 1]))
                        and not (m._forward_hooks or m._forward_pre_hooks or m.conv_branch._forward_hooks or m.i_branch._forward_hooks))
            else:
                mods = list(m._modules.values())
                cls = spconv.SparseConv3d if kind == "down" else spconv.SparseInverseConv3d
                good = (len(mods) == 3 and _plain_bn(mods[0]) and type(mods[1]) is nn.ReLU and _plain_conv(mods[2], cls, [2, 2, 2])
                        and (kind == "up" or mods[2].stride == [2, 2, 2]) and not (m._forward_hooks or m._forward_pre_hooks))
            ok[0] = ok[0] and good
        plan = self.__dict__["_doda_coarse"] = steps if ok[0] else False
        return plan

    def _forward_coarse(self, input):
        """The whole subtree in one executor call, or None when a precondition fails (then the modules run one by one)."""
        ext = Fsp._ext
        feats = input.features
        if (ext is None or not Fsp._SERIAL or not hasattr(ext, "coarse_ublock") or not feats.is_cuda
                or feats.dtype not in (torch.bfloat16, torch.float32) or feats.dim() != 2
                or _torch_module._global_forward_hooks or _torch_module._global_forward_pre_hooks
                or _torch_module._global_backward_hooks or _torch_module._global_backward_pre_hooks):
            return None
        backend = choose_coarse_backend(feats.shape[0], feats.dtype)
        if backend is None:
            return None
        plan = self._coarse_modules()
        if plan is False:
            return None
        training = self.training
        grad = torch.is_grad_enabled() and feats.requires_grad
        # per call (the plan is cached): a hook registered on ANY module of the subtree since then (feature taps, profilers) must
        # fire, and a frozen parameter must not receive a gradient — both mean module by module
        flat = self.__dict__.get("_doda_coarse_flat")
        if flat is None:
            flat = self.__dict__["_doda_coarse_flat"] = (list(self.modules()), list(self.parameters()))
        for m in flat[0]:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return None
        if grad:
            for q in flat[1]:
                if not q.requires_grad:
                    return None
        if grad and not (training and ext.get_defer_wgrad() and ext.get_direct_grads()):
            return None
        idict = input.indice_dict
        kinds, tensors, scalars = [], [], []
        gate_step = -1          # first step of level COARSE_LEVEL: where the rulebook prefetcher's gate opens (see forward)
        rows = {self.level: feats.shape[0]}
        for kind, m, lvl in plan:
            if kind == "rb":
                mods = list(m.conv_branch._modules.values())
                bn1, c1, bn2, c2 = mods[0], mods[2], mods[3], mods[5]
                data = idict.get(c1.indice_key)
                if (data is None or data.kind != "subm" or data.tbl.shape[1] != rows.get(lvl) or bn1.training != training
                        or bn2.training != training):
                    return None
                pk1, pk2 = c1._packed(feats, idict), c2._packed(feats, idict)
                if pk1 is None or pk2 is None:
                    return None
                sk = m.i_branch[0]
                if type(sk) is nn.Identity:
                    skt = [None, None, None, None, None]
                else:
                    pks = sk._packed(feats, idict)
                    if pks is None:
                        return None
                    ident = _cv._identity_table(rows[lvl], feats.device)
                    # (t[21]: the identity table doubles as both pair lists of the skip conv's weight gradient at the large levels)
                    skt = [sk._parameters["weight"], pks[0], pks[1], ident,
                           ident if (grad and Fsp._want_pairs(feats, sk._parameters["weight"], rows[lvl])) else None]
                if kind == "rb" and lvl == COARSE_LEVEL and gate_step < 0:
                    gate_step = len(kinds)
                kinds.append(0)
                tensors.append([data.tbl] + _bn_list(bn1) + [c1._parameters["weight"], pk1[0], pk1[1]] + _bn_list(bn2)
                               + [c2._parameters["weight"], pk2[0], pk2[1]] + skt)
                scalars.append([bn1.eps, bn1.momentum, bn2.eps, bn2.momentum])
            else:
                mods = list(m._modules.values())
                bn, cv = mods[0], mods[2]
                data = idict.get(cv.indice_key)
                if data is None or data.kind != "down2" or bn.training != training or data.tbl_rev.shape[1] != rows.get(lvl):
                    return None
                pk = cv._packed(feats, idict)
                if pk is None:
                    return None
                if kind == "down":
                    rows[lvl + 1] = data.outids.shape[0]
                    if rows[lvl + 1] < 2:
                        return None
                    kinds.append(1)
                    tensors.append([data.tbl, data.tbl_rev] + _bn_list(bn) + [cv._parameters["weight"], pk[0], pk[1]])
                    scalars.append([bn.eps, bn.momentum, rows[lvl + 1]])
                else:
                    kinds.append(2)
                    tensors.append([data.tbl_rev, data.tbl] + _bn_list(bn) + [cv._parameters["weight"], pk[0], pk[1]])
                    scalars.append([bn.eps, bn.momentum, rows[lvl]])
        st = input.__dict__.get("_doda_stats")
        stats_in = st[1] if (st is not None and st[0] is feats and st[2] == feats._version and torch.is_tensor(st[1])) else None
        gate = None
        if self.level < COARSE_LEVEL and gate_step >= 0 and _coarse_hooks and training and torch.is_grad_enabled():
            hooks = list(_coarse_hooks)

            def gate():
                for hook in hooks:
                    hook()
        y, stats = ext.coarse_ublock(feats, stats_in, kinds, tensors, scalars, training, gate_step if gate is not None else -1, gate)
        out = spconv.SparseConvTensor(y, input.indices, input.spatial_shape, input.batch_size)
        out.indice_dict = idict
        out.grid = input.grid
        if stats is not None:
            out._doda_stats = (y, stats, y._version)
        return out

    def forward(self, input):
        if self.level == COARSE_LEVEL and _coarse_hooks and self.training and torch.is_grad_enabled():   # (training steps only)
            for hook in list(_coarse_hooks):   # the step enters its coarse levels: few rows, most CUs idle from here on
                hook()
        if (self.level == EARLY_LEVEL and _early_exchange[0] is not None and self.training and torch.is_grad_enabled()
                and input.features.requires_grad):
            def _fire(g, fn=_early_exchange[0]):
                fn()
                return g
            input.features.register_hook(_fire)
        if COARSE_MODE != "off" and self.level == _coarse_entry_level():
            out = self._forward_coarse(input)
            if out is not None:
                return out
        out = self.blocks(input)
        if len(self.nPlanes) > 1:
            st_a = out.__dict__.get("_doda_stats")
            src = out.features                        # (what the level's statistics rows belong to)
            down = self._down_with_skip(out) if SKIP_VIA_BN else None
            if down is not None:
                skip_feats, down = down
            else:
                skip_feats = _shallow(out).features
                down = self.conv(out)                 # (re-binds out.features)
            dec = self.deconv(self.u(down))
            out.features = torch.cat((skip_feats, dec.features), dim=1)
            # BatchNorm statistics are per channel: the concatenation's are its halves' — the rows the two producing convs
            # accumulated in their epilogues — so the first BatchNorm of blocks_tail needs no sweep over the new tensor
            st_b = dec.__dict__.get("_doda_stats")
            out.__dict__.pop("_doda_stats", None)
            if (CAT_STATS and st_a is not None and st_b is not None and st_a[0] is src and st_a[2] == src._version
                    and st_b[0] is dec.features and st_b[2] == dec.features._version
                    and torch.is_tensor(st_a[1]) and torch.is_tensor(st_b[1])):
                out._doda_stats = (out.features, (st_a[1], st_b[1]), out.features._version)
            out = self.blocks_tail(out)
        if (self.level == WGRAD_SIDE_LEVEL and self.training and torch.is_grad_enabled() and out.features.requires_grad
                and out.features.is_cuda):
            out.features.register_hook(_side_flush_hook)
        return out

    def _down_with_skip(self, out):
        """self.conv = [BatchNorm, ReLU, strided conv] with the BatchNorm's pass-through output: returns (alias of
        out.features for the skip connection, the strided conv's output), or None when the sequence is not that plain
        triple in training mode on the fused path.  The concatenation's gradient for the skip — a column slice of a
        twice as wide matrix — then reaches the BatchNorm's backward kernel as its strided `add` operand."""
        from . import nn as _dnn
        from .spconv.modules import _run
        seq = self.conv
        mods = list(seq._modules.values())
        feats = out.features
        if (len(mods) != 3 or type(mods[1]) is not nn.ReLU or not isinstance(mods[2], spconv.SparseConvolution)
                or not mods[0].training or not torch.is_grad_enabled() or not feats.requires_grad
                or out.indices.shape[0] < 2 or not _dnn.fusable(mods[0], feats) or _dnn._ext is None
                or seq._forward_hooks or seq._forward_pre_hooks or seq._backward_hooks or seq._backward_pre_hooks
                or mods[0]._forward_hooks or mods[0]._forward_pre_hooks or mods[0]._backward_hooks
                or mods[1]._forward_hooks or mods[1]._forward_pre_hooks or mods[1]._backward_hooks):
            return None
        st = out.__dict__.get("_doda_stats")
        stats = st[1] if (st is not None and st[0] is feats and st[2] == feats._version) else None
        y, alias = _dnn.batch_norm_relu(feats, mods[0], True, True, stats)
        t = spconv.SparseConvTensor(y, out.indices, out.spatial_shape, out.batch_size)
        t.indice_dict = out.indice_dict
        t.grid = out.grid
        return alias, _run(mods[2], t)


class _PointLinear(Function):
    """scores[n] = feats[p2v[n]] @ W^T + b: the voxel->point gather and the Linear head of reference
    model/unet.py:62-64 as ONE gather-GEMM (K = 1 table = p2v).  Backward re-uses the same native
    kernels: d_feats[v] = sum over the voxel's points (table = transposed v2p map) of d_scores @ W,
    dW = wgrad over the p2v table."""

    @staticmethod
    def forward(ctx, feats, weight, bias, p2v, v2p_t):
        n = p2v.shape[0]
        fused_bias = (bias is not None and bias.dtype == torch.float32 and bias.is_cuda and feats.shape[1] % 4 == 0
                      and weight.shape[0] % 4 == 0)
        # (the bias rides in the gather kernel's store as a broadcast residual row: no separate pass over the scores)
        scores = None
        if fused_bias:
            try:
                scores = _ops.spconv_gather(feats.contiguous(), weight.view(1, *weight.shape), p2v.view(1, n), n, 1,
                                            weight.shape[0], out_f32=True, residual=bias.detach().contiguous(),
                                            residual_bcast=True)
            except _ops.DodaNativeError:   # (a shape the dense-table fast kernel does not take)
                fused_bias = False
        if scores is None:
            scores = _ops.spconv_gather(feats.contiguous(), weight.view(1, *weight.shape), p2v.view(1, n),
                                        n, 1, weight.shape[0], out_f32=True)
        if bias is not None and not fused_bias:
            scores += bias
        ctx.save_for_backward(feats, weight, p2v, v2p_t)
        ctx.has_bias = bias is not None
        return scores

    @staticmethod
    def backward(ctx, d_scores):
        feats, weight, p2v, v2p_t = ctx.saved_tensors
        n = p2v.shape[0]
        d_feats = d_w = d_b = None
        if (d_scores.is_cuda and d_scores.dtype == torch.float32 and feats.dtype == torch.bfloat16 and d_scores.dim() == 2
                and d_scores.shape[1] <= 64 and d_scores.shape[0] > 0):
            # bf16 operand of the two gather-GEMMs below and the column sums (d_bias) in ONE pass over the fp32 gradient
            dy, colsum = _ops.cast_colsum(d_scores.contiguous())
            if ctx.has_bias and ctx.needs_input_grad[2]:
                d_b = colsum
        else:
            dy = d_scores.contiguous().to(feats.dtype)
        if ctx.needs_input_grad[0]:
            k = v2p_t.shape[0]
            w_rep = weight.unsqueeze(0).expand(k, *weight.shape).contiguous()
            d_feats = _ops.spconv_gather(dy, w_rep, v2p_t, feats.shape[0], 0, weight.shape[1])
        if ctx.needs_input_grad[1]:
            d_w = _ops.spconv_wgrad(feats.contiguous(), dy, p2v.view(1, n), n)[0].t().to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2] and d_b is None:
            d_b = d_scores.sum(0)
        return d_feats, d_w, d_b, None, None


class _FusedCE(Function):
    @staticmethod
    def forward(ctx, scores, labels, ignore_index):
        scores = scores.contiguous()
        out, lse = _ops.cross_entropy_fwd(scores, labels, ignore_index)
        ctx.save_for_backward(scores, labels, lse, out)
        ctx.ignore_index = ignore_index
        return out[0]

    @staticmethod
    def backward(ctx, grad):
        scores, labels, lse, out = ctx.saved_tensors
        g = grad.reshape(1).to(torch.float32).contiguous()
        return _ops.cross_entropy_bwd(scores, labels, lse, out, g, ctx.ignore_index), None, None


class _VoxelHeadCE(Function):
    """loss = CrossEntropyLoss(ignore_index)(Linear(feats[p2v]), labels) (reference model/unet.py:62-64,107-108,196) computed at
    VOXEL level: every point of a voxel reads the same feature row, so the [points, classes] score matrix — written and re-read
    five times by the matrix path, ~210 us of the bench step — never exists (csrc/head.hip).  Second output: the per-voxel
    argmax class (the meters' prediction: pred_point = pred_voxel[p2v])."""

    @staticmethod
    def forward(ctx, feats, weight, bias, v2p, labels, ignore_index):
        out, pred = _ops.head_ce_fwd(feats, weight, bias, v2p, labels, ignore_index)
        ctx.save_for_backward(feats, weight, bias, v2p, labels, out)
        ctx.ignore_index = ignore_index
        ctx.mark_non_differentiable(pred)
        return out[0], pred

    @staticmethod
    def backward(ctx, grad, _grad_pred):
        feats, weight, bias, v2p, labels, out = ctx.saved_tensors
        g = grad.reshape(1).to(torch.float32)
        d_feats, dz, d_b = _ops.head_ce_bwd(feats, weight, bias, v2p, labels, ctx.ignore_index, out, g)
        d_w = None
        if ctx.needs_input_grad[1]:   # dW = dz^T feats (voxels as the MFMA k dimension: doda_head_dw_bf16; fp32: identity-table wgrad)
            d_w = _ops.head_dw(feats, dz).to(weight.dtype)
        return (d_feats if ctx.needs_input_grad[0] else None), d_w, (d_b if bias is not None and ctx.needs_input_grad[2] else None), \
            None, None, None


FUSED_HEAD_LOSS = _os.environ.get("DODA_FUSED_HEAD_LOSS", "1") == "1"


def cross_entropy(scores, labels, ignore_index=255):
    """nn.CrossEntropyLoss(ignore_index) (reference model/unet.py:108,196).  Device fp32 logits with at
    most 64 classes go through the fused native pair (doda_cross_entropy_fwd/_bwd: 3 launches instead
    of ~15 torch kernels that each stream the [N, C] matrix); anything else through log_softmax + gather
    (torch's own nll_loss reduction kernels take ~1.6 ms per step at 800k points x 20 classes)."""
    if (scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.shape[1] <= 64
            and labels.dtype == torch.int64 and scores.shape[0] > 0):
        return _FusedCE.apply(scores, labels.contiguous(), int(ignore_index))
    valid = labels != ignore_index
    logp = torch.log_softmax(scores.float(), dim=1)
    picked = logp.gather(1, labels.clamp(0, scores.shape[1] - 1).unsqueeze(1)).squeeze(1)
    return -(picked * valid).sum() / valid.sum().clamp(min=1)


class SparseConvNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        bb = cfg.MODEL.BACKBONE
        try:
            n_classes = cfg.COMMON_CLASSES.n_classes
        except AttributeError:
            n_classes = cfg.DATA_CONFIG.DATA_CLASS.n_classes
        m = bb.mid_channel
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        block = ResidualBlock if bb.block_residual else VGGBlock
        self.input_conv = spconv.SparseSequential(_subm3(bb.in_channel, m, "subm1"))
        self.unet = UBlock([m * i for i in range(1, 8)], norm_fn, bb.block_reps, block, indice_key_id=1)
        self.output_layer = spconv.SparseSequential(norm_fn(m), nn.ReLU())
        self.linear = nn.Linear(m, n_classes)
        for mod in self.modules():  # reference set_bn_init (unet.py:51-56)
            if "BatchNorm" in mod.__class__.__name__:
                mod.weight.data.fill_(1.0)
                mod.bias.data.fill_(0.0)

    def head_loss(self, feats, v2p_map, labels, ignore_index=255):
        """Linear head + CrossEntropyLoss on the voxel features without the point-level score matrix (_VoxelHeadCE), or None
        when that form does not apply (then: scores = forward(...), cross_entropy(scores, labels)).  Sets self.voxel_pred."""
        if not (FUSED_HEAD_LOSS and feats.is_cuda and feats.dim() == 2 and feats.shape[1] == 16 and feats.shape[0] > 0
                and feats.dtype in (torch.float32, torch.bfloat16) and self.linear.out_features <= 32
                and self.linear.weight.dtype == torch.float32 and v2p_map is not None and v2p_map.is_cuda
                and v2p_map.dtype == torch.int32 and v2p_map.dim() == 2 and v2p_map.shape[0] == feats.shape[0]
                and labels.is_cuda and labels.dtype == torch.int64
                and not (self.linear._forward_hooks or self.linear._forward_pre_hooks or self.linear._backward_hooks)):
            return None
        loss, pred = _VoxelHeadCE.apply(feats, self.linear.weight, self.linear.bias, v2p_map, labels, int(ignore_index))
        self.voxel_pred = pred
        return loss

    def forward(self, input, input_map, return_mid_feat=False, v2p_map=None, v2p_map_t=None, labels=None, ignore_index=255):
        if input.features.is_cuda and input.indices.shape[0] > 0:
            # all 13 rulebooks up front (+ their pair lists when a bf16 backward pass will follow)
            spconv.ops.build_pyramid(input, len(self.unet.nPlanes), with_pairs=(
                spconv.functional.WGRAD_PAIRS and torch.is_grad_enabled() and self.training
                and input.features.dtype == torch.bfloat16),
                with_tiles=tile_levels_for(input.features.dtype))   # tilebooks serve inference as well
        out = self.output_layer(self.unet(self.input_conv(input)))
        feats = out.features
        if labels is not None and not return_mid_feat:      # the caller wants the loss: head + loss at voxel level
            loss = self.head_loss(feats, v2p_map, labels, ignore_index)
            if loss is not None:
                return loss
        fused = (v2p_map is not None and not return_mid_feat and feats.is_cuda
                 and input_map.dtype == torch.int32 and 0 < v2p_map.shape[1] - 1 <= 27
                 and feats.shape[1] % 4 == 0 and self.linear.out_features % 4 == 0)
        if fused:  # voxel->point gather + Linear as one gather-GEMM over the p2v table
            v2p_t = v2p_map_t if (v2p_map_t is not None and v2p_map_t.shape == (v2p_map.shape[1] - 1, v2p_map.shape[0])
                                  and v2p_map_t.is_contiguous()) else v2p_map[:, 1:].t().contiguous()
            scores = _PointLinear.apply(feats, self.linear.weight, self.linear.bias, input_map, v2p_t)
            if labels is not None:
                self.voxel_pred = None
                self.point_scores = scores
                return cross_entropy(scores, labels, ignore_index)
            return scores
        point_feats = feats[input_map.long()]  # voxel -> point
        scores = self.linear(point_feats.to(self.linear.weight.dtype))
        if labels is not None and not return_mid_feat:
            self.voxel_pred = None
            self.point_scores = scores
            return cross_entropy(scores, labels, ignore_index)
        return (point_feats, scores) if return_mid_feat else scores


_PYR_STREAMS = {}


F32_TILES = _os.environ.get("DODA_F32_TILES", "0") == "1"


# bf16 levels with a tilebook (DODA_TILE_LEVELS).  Levels 1-2 use it for the gathers AND the weight gradient; a level-3 tilebook
# (48 channels: no tile gather kernel) would serve the weight gradient only.
BF16_TILE_LEVELS = int(_os.environ.get("DODA_TILE_LEVELS", "2"))


def tile_levels_for(dtype):
    """Levels whose SubM rulebook gets a tilebook: the two finest for bf16 (rows of 32 / 64 bytes).  The tile kernel also
    takes fp32 16-channel rows (DODA_F32_TILES=1: level 1), but fp32 layers are bound by the fp32 matrix rate, 1/16 of
    bf16: measured in round 4 the fp32 step takes 12.2 ms with the tilebook against 11.9 ms on the dense-table kernels
    (conv_tile MODE 2 96 us against conv_fast 97 us per level-1 layer, and the tilebook build on top): off by default."""
    if dtype == torch.bfloat16:
        return BF16_TILE_LEVELS
    return 1 if (dtype == torch.float32 and F32_TILES) else 0


def _prebuild_pyramid(voxel_coords, spatial_shape, batch_size, n_levels, device, with_pairs=False, with_tiles=None):
    """Build the int32 indices and all rulebooks on a side stream that does NOT wait for the main
    stream.  Rulebooks depend only on the voxel coordinates; built at the head of the forward pass on
    the main stream, each of their six size read-backs blocks the host until the previous step's
    whole backward has drained, and the GPU then idles while the host catches up (measured: 1.1 ms of
    wall time per step for 0.55 ms of kernels).  Only valid when `voxel_coords` has no producer work
    pending on any stream (a resident batch) — the caller opts in.
    Returns (indices int32, indice_dict); every tensor is handed over to the main stream."""
    main = torch.cuda.current_stream(device)
    side = _PYR_STREAMS.get(device)
    if side is None:
        from .streams import independent_stream
        side = _PYR_STREAMS[device] = independent_stream(device, tag="rulebooks")
    with torch.cuda.stream(side):
        idx32 = voxel_coords.int()
        probe = spconv.SparseConvTensor(None, idx32, spatial_shape, batch_size)
        spconv.ops.build_pyramid(probe, n_levels, with_pairs=with_pairs, with_tiles=with_tiles)
    main.wait_stream(side)
    idx32.record_stream(main)
    for data in probe.indice_dict.values():
        for t in vars(data).values():
            for u in (t if isinstance(t, tuple) else (t,)):
                if torch.is_tensor(u) and u.is_cuda:
                    u.record_stream(main)
    return idx32, probe.indice_dict


class PyramidPrefetcher:
    """Rulebooks as part of the data pipeline: the 13 rulebooks of batch k+1 are built on a helper thread
    and a side stream while the main thread issues step k.

    Rulebooks depend only on voxel coordinates — in the reference they are built inside spconv's forward,
    while its voxelisation already runs ahead of the step in DataLoader workers (dataset/dataset.py:121-187).
    Built in line, the six size read-backs of a pyramid (one per strided level) block the issuing thread
    for ~1.2 ms per step (tools/hostprof.py), and the step is paced by host issue time as much as by the
    GPU.  Here they block the helper thread instead (the extension call releases the GIL); the consumer
    picks the finished pyramid up with one stream-wait.  Every batch still gets its own rulebooks: nothing
    is cached across batches."""

    def __init__(self, device, n_levels, gated=False):
        """gated: the side stream does not start a build before the main stream's forward pass has reached the coarse
        levels (COARSE_LEVEL).  Measured: the same step with a re-used pyramid runs at 5.7 ms, with the next pyramid
        built from the start of the step at 6.5 ms — 0.9 ms of rulebook kernels next to the level-1 convolutions cost
        0.8 ms; next to the coarse levels' launch-floor-bound kernels they are nearly free."""
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.device, self.n_levels = device, n_levels
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="doda-rulebooks")
        from .streams import independent_stream
        self.stream = independent_stream(device, tag="rulebooks")   # (not on the main stream's hardware queue)
        self.gated = bool(gated)
        self._gate = None            # (threading.Event, torch.cuda.Event) of the build waiting for this step's coarse phase
        self._lock = threading.Lock()
        self.gate_timeouts = 0       # builds that started WITHOUT the coarse-phase event (steps longer than the wait)
        if self.gated:
            _coarse_hooks.append(self._open_gate)

    def _open_gate(self):
        """Called from UBlock.forward at COARSE_LEVEL (main thread, main stream)."""
        with self._lock:
            gate, self._gate = self._gate, None
        if gate is not None:
            gate[1].record(torch.cuda.current_stream(self.device))
            gate[0].set()

    def submit(self, batch, with_pairs=False, with_tiles=None, resident=False, now=False):
        """now: build at once (the first batch: no step in flight whose coarse phase could open the gate).
        batch: collated dictionary whose `voxel_locs` is on the device.  Work still queued on the CALLER's
        current stream that produces it (collate_device returns with doda_voxelize_idx_fill pending) is ordered
        before the build through an event recorded here; resident=True (a batch that was complete before the
        call, e.g. bench.py's reused one) skips that event, so the build does not queue behind the step in flight."""
        import threading
        coords, shape = batch["voxel_locs"], batch["spatial_shape"]
        bs = batch["offsets"].numel() - 1
        ready = None
        if coords.is_cuda and not resident:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        gate = None
        if self.gated and not now:
            gate = (threading.Event(), torch.cuda.Event())
            with self._lock:
                stale, self._gate = self._gate, gate
            if stale is not None:      # a build whose step never reached the coarse levels: let it go
                stale[0].set()
        return self.pool.submit(self._build, coords, shape, bs, with_pairs, with_tiles, ready, gate)

    def _build(self, coords, shape, bs, with_pairs, with_tiles=None, ready=None, gate=None):
        torch.cuda.set_device(self.device)
        if gate is not None:
            # wait (host side) until the main thread has recorded the coarse-phase event of the step in flight; a step
            # that never gets there (evaluation of a shallow model, an exception) must not hold the pipeline: time out
            opened = gate[0].wait(timeout=0.05)
            if not opened:
                # the step in flight did not reach its coarse levels in time (first iterations, evaluation, a profiler
                # run): this build goes ungated; the stale gate is withdrawn so that nobody records an event for it, and
                # the count tells an A/B reader how many of its steps were NOT gated
                with self._lock:
                    if self._gate is gate:
                        self._gate = None
                    self.gate_timeouts += 1
        with torch.cuda.stream(self.stream):
            if gate is not None and opened:
                self.stream.wait_event(gate[1])
            if ready is not None:
                self.stream.wait_event(ready)
                coords.record_stream(self.stream)   # read here; the allocator must not recycle it under the build
            idx32 = coords.int()
            probe = spconv.SparseConvTensor(None, idx32, shape, bs)
            spconv.ops.build_pyramid(probe, self.n_levels, with_pairs=with_pairs, with_tiles=with_tiles)
            done = torch.cuda.Event()
            done.record(self.stream)
        return idx32, probe.indice_dict, done

    @staticmethod
    def take(future, device):
        """(indices int32, indice_dict) of a submitted batch, handed over to the current stream."""
        idx32, book, done = future.result()
        main = torch.cuda.current_stream(device)
        main.wait_event(done)
        idx32.record_stream(main)
        for data in book.values():
            for t in vars(data).values():
                for u in (t if isinstance(t, tuple) else (t,)):
                    if torch.is_tensor(u) and u.is_cuda:
                        u.record_stream(main)
        return idx32, book

    def shutdown(self):
        with self._lock:
            gate, self._gate = self._gate, None
        if gate is not None:
            gate[0].set()
        self.pool.shutdown(wait=True)
        if self.gated and self._open_gate in _coarse_hooks:
            _coarse_hooks.remove(self._open_gate)


INPUT_ROWS = _os.environ.get("DODA_INPUT_ROWS", "1") == "1"


def _input_rows(net, feats, xyz, v2p, mode, feature_dtype):
    """The input layer's rows in one launch (ops.voxelize_fp_rows: cat + voxel pooling + cast + the layer's channel padding,
    reference model/unet.py:89-94), or None when that form does not apply (CPU, features with a gradient, another input layer)."""
    if not (INPUT_ROWS and feats.is_cuda and v2p.is_cuda and not feats.requires_grad and (xyz is None or not xyz.requires_grad)
            and feats.dtype == torch.float32 and (xyz is None or xyz.dtype == torch.float32) and v2p.dtype == torch.int32
            and v2p.dim() == 2 and v2p.shape[0] > 0 and feature_dtype in (torch.float32, torch.bfloat16) and mode in (3, 4)):
        return None
    conv = getattr(net, "input_conv", None)
    conv = conv[0] if conv is not None and len(conv) == 1 else None
    if not isinstance(conv, spconv.SubMConv3d):
        return None
    c_in = feats.shape[1] + (0 if xyz is None else xyz.shape[1])
    c_out = _cv.padded_in_channels(conv, feature_dtype)
    if conv.in_channels != c_in:
        return None
    rows = _ops.voxelize_fp_rows(feats.contiguous(), None if xyz is None else xyz.contiguous(), v2p.contiguous(), mode,
                                  c_out or c_in, feature_dtype)
    if c_out:
        rows._doda_padded_from = c_in
    return rows


def point_predictions(model, p2v):
    """argmax class per POINT after a `labels=` call of voxelize_and_run: the per-voxel argmax of the fused head gathered through
    p2v, or the argmax of the score matrix when the matrix path ran."""
    net = model.module if hasattr(model, "module") else model
    vp = getattr(net, "voxel_pred", None)
    if vp is not None:
        return vp[p2v.long()].long()
    return net.point_scores.detach().argmax(1)


def voxelize_and_run(cfg, model, batch, device, feature_dtype=torch.float32, fused_head=True,
                     inputs_ready=False, pyramid=None, labels=None, ignore_index=255):
    """reference model/unet.py:72-99 (test_model_feat): H2D, voxel mean-pooling, network.
    labels: return the training LOSS instead of the per-point scores (reference model/unet.py:107-108,196: CrossEntropyLoss on the
    head's scores) — computed at voxel level without the score matrix when the head allows it; point_predictions() afterwards
    gives the argmax per point for the meters.
    inputs_ready: the batch is resident on `device` with no copy or kernel still producing it, so the
    rulebooks may be built on a side stream ahead of the main stream's queue (_prebuild_pyramid).
    pyramid: (indices int32, indice_dict) of this batch from PyramidPrefetcher.take."""
    voxel_coords = batch["voxel_locs"].to(device, non_blocking=True)
    p2v = batch["p2v_map"].to(device, non_blocking=True)
    v2p = batch["v2p_map"].to(device, non_blocking=True)
    feats = batch["feats"].to(device, non_blocking=True)
    xyz = batch["locs_float"].to(device, non_blocking=True) if cfg.MODEL.BACKBONE.use_xyz else None
    batch_size = batch["offsets"].numel() - 1
    net = model.module if hasattr(model, "module") else model
    voxel_feats = _input_rows(net, feats, xyz, v2p, cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_mode, feature_dtype)
    if voxel_feats is None:
        if xyz is not None:
            feats = torch.cat((feats, xyz), 1)
        voxel_feats = pointgroup_ops.voxelization(feats, v2p, cfg.DATA_CONFIG.DATA_PROCESSOR.voxel_mode)
    if pyramid is not None:
        idx32, book = pyramid
        inp = spconv.SparseConvTensor(voxel_feats.to(feature_dtype), idx32, batch["spatial_shape"], batch_size)
        inp.indice_dict.update(book)
    elif (inputs_ready and batch["voxel_locs"].is_cuda and voxel_coords.shape[0] > 0
            and hasattr(net, "unet") and device.type == "cuda"):
        idx32, pyramid = _prebuild_pyramid(voxel_coords, batch["spatial_shape"], batch_size,
                                           len(net.unet.nPlanes), device,
                                           with_pairs=(spconv.functional.WGRAD_PAIRS and torch.is_grad_enabled()
                                                       and net.training and feature_dtype == torch.bfloat16),
                                           with_tiles=tile_levels_for(feature_dtype))
        inp = spconv.SparseConvTensor(voxel_feats.to(feature_dtype), idx32, batch["spatial_shape"], batch_size)
        inp.indice_dict.update(pyramid)
    else:
        inp = spconv.SparseConvTensor(voxel_feats.to(feature_dtype), voxel_coords.int(),
                                      batch["spatial_shape"], batch_size)
    if fused_head:
        v2p_t = batch.get("v2p_map_t")
        if labels is not None:   # -> the LOSS (head + CrossEntropyLoss at voxel level when the fused form applies)
            return model(inp, p2v, v2p_map=v2p, v2p_map_t=v2p_t.to(device, non_blocking=True) if v2p_t is not None else None,
                         labels=labels, ignore_index=ignore_index)
        return model(inp, p2v, v2p_map=v2p, v2p_map_t=v2p_t.to(device, non_blocking=True) if v2p_t is not None else None)
    if labels is not None:
        return cross_entropy(model(inp, p2v), labels, ignore_index)
    return model(inp, p2v)
