"""SubMConv3d / SparseConv3d / SparseInverseConv3d (spconv v1.2 `spconv.conv`).

Constructor arguments, the `[kD,kH,kW,Cin,Cout]` weight Parameter (state-dict compatible with
DODA's model zoo), kaiming_uniform(a=sqrt(5)) initialisation, rulebook caching by
`indice_key` in `input.indice_dict`, and the 1x1 shortcut (`features @ W.view(Cin,Cout)`) follow
upstream; the arithmetic is libdoda_hip.so's."""
import math
import weakref

import torch
from torch import nn
from torch.nn import init

from .. import ops as _nops
from . import functional as Fsp
from . import ops
from .core import IndiceDict, SparseConvTensor
from .modules import SparseModule

# ---------------------------------------------------------------------------------------------
# Weight pre-packing.  The native conv kernels read weights in MFMA-fragment order (bf16-rounded for
# bf16 features; transposed / offset-mirrored for the data-grad).  Instead of packing inside every
# conv call (2 x 71 launches per U-Net step) all live convolution modules of a device are re-packed
# by ONE multi-tensor launch (~20 us) whenever the weights may have changed.
#
# "May have changed" must not depend on something an optimizer can bypass: `weight._version` is NOT
# bumped by fused optimizers (torch 2.10: SGD/Adam(fused=True) leave it untouched) nor by edits through
# `.data`.  The packed copies therefore belong to a GENERATION, and a new generation starts
#   * with every new forward pass: the first convolution that sees a SparseConvTensor whose
#     `indice_dict` carries no generation token starts one (a network input is a fresh tensor each
#     step, and every tensor derived from it shares its indice_dict);
#   * after every `optimizer.step()` of any torch optimizer (global step post-hook);
#   * on `invalidate_packed()`, `load_state_dict`, `.to()/.float()/...` (Module._apply).
# Within a generation `_version` and `data_ptr()` are still compared, which catches ordinary in-place
# edits between two calls on the SAME input tensor; only a `.data` edit between two such calls needs
# an explicit invalidate_packed().
# The cache holds weak references only: a deleted model drops out of the plan.
# ---------------------------------------------------------------------------------------------
import os as _os

_MODULES = weakref.WeakSet()
_PLANS = {}   # (device, elem_bytes) -> _Plan
_PREPACK = _os.environ.get("DODA_NO_PREPACK", "0") != "1"
_GEN = [1]    # current weight generation
_PAD_INPUT_16 = _os.environ.get("DODA_PAD_INPUT16", "1") == "1"


def padded_in_channels(conv, dtype):
    """Channels the rows of `conv`'s input are zero-padded to on the GPU (forward below), or None when it takes them as they are."""
    # (only where forward below pads: a 3x3x3 SubM / strided layer — the 1x1 shortcut multiplies [N, c_in] x [c_in, c_out] with
    # torch.mm and would meet pre-padded rows with a shape mismatch)
    if conv.in_channels % 4 == 0 or getattr(conv, "conv1x1", False) or list(conv.kernel_size) != [3, 3, 3]:
        return None
    extra = (-conv.in_channels) % 4
    if _PAD_INPUT_16 and dtype == torch.bfloat16 and conv.in_channels < 16 and conv.subm and conv.kernel_size == [3, 3, 3]:
        extra = 16 - conv.in_channels
    return conv.in_channels + extra


def set_prepack(on):
    """Switch the one-launch pre-pack on or off at run time (off: every conv call packs its own
    weights inside the call — the reference behaviour for the parity tests)."""
    global _PREPACK
    _PREPACK = bool(on)
    invalidate_packed()
    return _PREPACK


def new_generation():
    """Declare every packed weight copy stale (cheap: the re-pack happens lazily, once, at the next
    convolution call)."""
    _GEN[0] += 1
    return _GEN[0]


def invalidate_packed():
    new_generation()
    _PLANS.clear()
    for m in list(_MODULES):
        m._doda_packed = {}


def _after_optimizer_step(optimizer, args, kwargs):
    new_generation()


try:   # any torch optimizer, fused or not (reference tool/train.py:100-104 plain optimizer.step())
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post
    _reg_post(_after_optimizer_step)
except ImportError:   # pragma: no cover  (torch < 2.0)
    pass


def _enter_forward_pass(indice_dict):
    """Start a new generation the first time a convolution meets the rulebook dictionary of a forward
    pass.  A foreign dictionary (user code assigned a plain dict) has no slot for the token: every call
    then starts a generation — slower, never stale."""
    if type(indice_dict) is IndiceDict:
        if indice_dict.pack_gen is None:
            indice_dict.pack_gen = new_generation()
    else:
        new_generation()


_IDENT = {}


def _identity_table(n, device):
    """[1, n] int32 table 0..n-1 for the K = 1 gather-GEMM of 1x1 convolutions: a prefix view of one
    grow-only arange per device (no launch per call)."""
    buf = _IDENT.get(device)
    if buf is None or buf.numel() < n:
        buf = _IDENT[device] = torch.arange(max(n, 1 << 20), dtype=torch.int32, device=device)
    return buf[:n].view(1, n)


def _bwd_layout(m):
    return 2 if (m.subm and not m.conv1x1) else 1


class _Plan:
    """One pack launch for all convolution modules of (device, elem_bytes)."""
    __slots__ = ("sig", "plan", "refs", "gen", "versions")

    def __init__(self, sig, plan, mods):
        self.sig, self.plan = sig, plan
        self.refs = [weakref.ref(m) for m in mods]
        self.gen = -1
        self.versions = ()


def _repack_all(device, esz):
    # (once per forward pass over all 71 conv modules: one dictionary look-up per module instead of a handful of
    # nn.Module.__getattr__ calls each)
    cand = []
    for m in _MODULES:
        w = m._parameters.get("weight")
        if (w is not None and w.device == device and w.dtype == torch.float32 and w.is_contiguous()
                and w.shape[0] * w.shape[1] * w.shape[2] <= 27):
            cand.append((id(m), m, w))
    cand.sort(key=lambda t: t[0])
    mods = [t[1] for t in cand]
    sig = tuple((t[0], t[2].data_ptr()) for t in cand)
    cached = _PLANS.get((device, esz))
    if cached is None or cached.sig != sig:
        entries = []
        for m in mods:
            K = m.weight.shape[0] * m.weight.shape[1] * m.weight.shape[2]
            w = m.weight.detach().view(K, m.in_channels, m.out_channels)
            entries.append((w, K, m.in_channels, m.out_channels, 0, esz))          # forward: [K][kc][nc]
            entries.append((w, K, m.out_channels, m.in_channels, _bwd_layout(m), esz))  # data-grad
        cached = _PLANS[(device, esz)] = _Plan(sig, _nops.PackPlan(entries, device), mods)
        for k, m in enumerate(mods):
            m._doda_packed[esz] = (cached, k, m._parameters["weight"], m.weight.data_ptr())
    cached.plan.run()
    cached.gen = _GEN[0]
    cached.versions = tuple(t[2]._version for t in cand)
    return cached


class SparseConvolution(SparseModule):
    supports_residual = True   # forward(input, residual=...) adds a feature matrix to the output

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False, use_hash=False):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("groups != 1")
        if ndim != 3:
            raise NotImplementedError("doda_amd implements 3-D sparse convolution only")
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = ops._triple(kernel_size)
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.stride = ops._triple(stride)
        self.padding = ops._triple(padding)
        self.dilation = ops._triple(dilation)
        self.output_padding = ops._triple(output_padding)
        self.transposed = transposed
        self.inverse = inverse
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.use_hash = use_hash
        if inverse and indice_key is None:
            raise ValueError("SparseInverseConv3d needs the indice_key of its strided convolution")
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        self._doda_packed = {}
        _MODULES.add(self)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "{}, {}, kernel_size={}, stride={}, subm={}, inverse={}, indice_key={}".format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.subm,
            self.inverse, self.indice_key)

    def _packed(self, features, indice_dict=None):
        """(forward, data-grad) fragment-packed weights for this feature dtype, or None when the
        native fast path does not apply.  indice_dict: the input tensor's rulebook dictionary, which
        carries the generation token of the forward pass (see the comment at the top of this file)."""
        if indice_dict is not None and getattr(indice_dict, "pack_gen", None) is None:
            _enter_forward_pass(indice_dict)
        # hot path first (called for every conv of every step; the step is issue-bound on the host):
        # current generation, same Parameter object, same version, same storage -> the cached pair
        w = self._parameters["weight"]
        esz = 4 if features.dtype == torch.float32 else 2
        st = self._doda_packed.get(esz)
        if st is not None:
            plan, k = st[0], st[1]
            if (plan.gen == _GEN[0] and st[2] is w and st[3] == w.data_ptr()
                    and plan.versions[k] == w._version):
                out = plan.plan.outputs
                return out[2 * k], out[2 * k + 1]
        if not (_PREPACK and features.is_cuda and w.is_cuda and w.dtype == torch.float32
                and features.dtype in (torch.float32, torch.bfloat16) and w.is_contiguous()):
            return None
        if w.shape[0] * w.shape[1] * w.shape[2] > 27:
            return None
        _MODULES.add(self)   # e.g. a deep-copied module never ran __init__
        plan = _repack_all(w.device, esz)   # rewrites every module's copies, under the current generation
        st = self._doda_packed.get(esz)
        if st is None or st[0] is not plan or st[2] is not w:
            return None
        out = plan.plan.outputs
        return out[2 * st[1]], out[2 * st[1] + 1]

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .float() / .bfloat16(): storage (and maybe dtype) of the weight changes
        self._doda_packed = {}
        new_generation()
        return super()._apply(fn, recurse)

    def _load_from_state_dict(self, *args, **kwargs):
        new_generation()
        return super()._load_from_state_dict(*args, **kwargs)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_doda_packed" else copy.deepcopy(v, memo)
        _MODULES.add(new)
        return new

    def forward(self, input, residual=None):
        """residual (extension over spconv): optional feature matrix added to the output, fused into the
        native kernel's store for 3x3x3 SubM convolutions (`y = conv(x) + residual`)."""
        assert isinstance(input, SparseConvTensor)
        features = input.features
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size

        if self.conv1x1:
            native = (features.is_cuda and features.dtype in (torch.float32, torch.bfloat16)
                      and self.in_channels % 4 == 0 and self.out_channels % 4 == 0
                      and self.weight.dtype == torch.float32 and features.shape[0] > 0)
            if native:
                ident = _identity_table(features.shape[0], features.device)
                # (no rulebook is involved, and the reference hands its 1x1 skip convolution a FRESH tensor
                # header, model/unet_block.py:33: an empty dictionary here is not the start of a forward pass)
                out_features = Fsp.conv1x1(features, self.weight, ident, self._packed(
                    features, input.indice_dict if len(input.indice_dict) else None), want_stats=self.training)
                out_features, stats = out_features if self.training else (out_features, None)
            else:
                w2 = self.weight.view(self.in_channels, self.out_channels)
                out_features = torch.mm(features, w2.to(features.dtype))
                stats = None
            if self.bias is not None:
                out_features = out_features + self.bias.to(features.dtype)
                stats = None
            if residual is not None:
                out_features = out_features + residual
                stats = None
            out = SparseConvTensor(out_features, indices, spatial_shape, batch_size)
            out.indice_dict = input.indice_dict
            out.grid = input.grid
            if stats is not None:
                out._doda_stats = (out_features, stats, out_features._version)
            return out

        weight, packed = self.weight, None
        if features.is_cuda and self.in_channels % 4 != 0:
            # e.g. the xyz input layer (3 channels): rows of 6 / 12 bytes miss the aligned vector
            # path of the native kernels (measured 107 us against ~30 us at 600k voxels).  Zero
            # channels appended to the features and zero rows to the weight leave the result
            # unchanged; autograd slices the weight gradient back.
            extra = (-self.in_channels) % 4
            if (_PAD_INPUT_16 and features.dtype == torch.bfloat16 and self.in_channels < 16 and self.subm
                    and self.kernel_size == [3, 3, 3]):
                # ... and up to 16 channels for bf16 SubM layers: 32-byte rows take the LDS-staged tile kernels over the
                # rulebook's tilebook (forward 69 -> 34 us, weight gradient 85 -> ~31 us at 600k voxels) instead of the
                # 8-byte-row generic paths, for one extra 19 MB tensor
                extra = 16 - self.in_channels
            if getattr(features, "_doda_padded_from", None) == self.in_channels and features.shape[1] == self.in_channels + extra:
                pass        # already this layer's padded rows (model.voxelize_and_run: doda_voxelize_fp_rows wrote them)
            elif features.is_cuda and not features.requires_grad and features.is_contiguous() and features.shape[0] > 0:
                features = _nops.pad_channels(features, self.in_channels + extra)   # one kernel (torch: fill + strided copy)
            else:
                features = nn.functional.pad(features, (0, extra))
            weight = nn.functional.pad(self.weight, (0, 0, 0, extra))
        else:
            packed = self._packed(features, input.indice_dict)

        data = input.find_indice_pair(self.indice_key)
        if self.inverse:
            if data is None or data.kind != "down2":
                raise RuntimeError("SparseInverseConv3d: no strided rulebook under indice_key %r"
                                   % (self.indice_key,))
            outids, out_spatial_shape = data.indices, data.spatial_shape
            out_features = Fsp.indice_inverse_conv(features, weight, data, packed, want_stats=self.training)
        else:
            if data is None:
                if self.subm:
                    data = ops.build_subm(indices, batch_size, spatial_shape, self.kernel_size)
                else:
                    data = ops.build_down2(indices, batch_size, spatial_shape, self.kernel_size,
                                           self.stride, self.padding, self.dilation)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = data
            outids, out_spatial_shape = data.outids, data.out_spatial_shape
            if self.subm:
                fuse = (residual is not None and features.is_cuda and residual.dtype == features.dtype
                        and tuple(residual.shape) == (outids.shape[0], self.out_channels))
                out_features = Fsp.indice_subm_conv(features, weight, data, packed, residual if fuse else None,
                                                    want_stats=self.training)
                if fuse:
                    residual = None
            else:
                out_features = Fsp.indice_conv(features, weight, data, packed, want_stats=self.training)

        # (training mode: the conv also returned the BatchNorm statistics partials of its output, or None)
        out_features, stats = out_features if self.training else (out_features, None)
        if residual is not None:
            out_features = out_features + residual
            stats = None
        if self.bias is not None:
            out_features = out_features + self.bias.to(out_features.dtype)
            stats = None
        out = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        if stats is not None:   # (sum y, sum y^2) partials for a fused BatchNorm applied to exactly these features
            # (... and to exactly this VERSION of them: `output.features += skip` of the reference's residual
            # block, model/unet_block.py:36, edits the same tensor object in place)
            out._doda_stats = (out_features, stats, out_features._version)
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, use_hash=False):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key, use_hash=use_hash)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, use_hash=False):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key, use_hash=use_hash)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)
