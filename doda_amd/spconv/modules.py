"""SparseModule marker base class and SparseSequential container (spconv v1.2 `spconv.modules`;
reference imports at model/unet_block.py:3-4, uses at model/unet.py:35,42 and
model/unet_block.py:10,41,62-85)."""
from collections import OrderedDict

from torch import nn
from torch.nn.modules import module as _mod

from .. import nn as _dnn
from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Modules deriving from this are called with the SparseConvTensor itself; anything else in
    a SparseSequential is applied to `.features`."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def _run(module, *args, **kwargs):
    """module(*args) without nn.Module.__call__'s hook machinery when the module has no hooks: a
    U-Net step makes ~200 child calls and the step is issue-bound on the host."""
    if (module._forward_hooks or module._forward_pre_hooks or module._backward_hooks
            or module._backward_pre_hooks or _mod._global_forward_hooks or _mod._global_forward_pre_hooks
            or _mod._global_backward_hooks or _mod._global_backward_pre_hooks):
        return module(*args, **kwargs)
    return module.forward(*args, **kwargs)


class SparseSequential(SparseModule):
    """nn.Sequential for mixed sparse / dense-feature modules.

    Accepts positional modules, a single OrderedDict, or keyword-named modules (upstream API).
    Non-sparse modules (BatchNorm1d, ReLU, Identity, DSNorm...) act on `.features`, which is
    re-bound on the same tensor object; they are skipped when the tensor has no active rows."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input, residual=None):
        """residual (extension over spconv): feature matrix added to the output of the sequence; when
        the last module is a sparse convolution the add is fused into its kernel.  residual="input" adds
        the sequence's own input features (identity skip of a residual block): if the sequence starts
        with a fused BatchNorm the skip's gradient is summed inside that BatchNorm's backward kernel."""
        mods = list(self._modules.values())
        k = 0
        take_input = isinstance(residual, str)
        if take_input:
            assert residual == "input"
            residual = input.features
        while k < len(mods):
            module = mods[k]
            k += 1
            if is_spconv_module(module):
                if residual is not None and k == len(mods) and getattr(module, "supports_residual", False):
                    input = _run(module, input, residual=residual)
                    residual = None
                else:
                    input = _run(module, input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if _dnn.fusable(module, input.features):
                        # BatchNorm1d [-> ReLU] on .features: one fused HIP path (doda_amd.nn); when the
                        # features came out of a conv that accumulated their statistics, those are used
                        relu = k < len(mods) and type(mods[k]) is nn.ReLU
                        st = input.__dict__.get("_doda_stats")
                        # the statistics belong to one tensor object AND one version of it: an in-place edit
                        # (`output.features += ...`, reference model/unet_block.py:36) keeps the object
                        feats = input.features
                        stats = st[1] if (st is not None and st[0] is feats and st[2] == feats._version) else None
                        if take_input and k == 1 and module.training:
                            input.features, residual = _dnn.batch_norm_relu(input.features, module, relu, True, stats)
                        else:
                            input.features = _dnn.batch_norm_relu(input.features, module, relu, False, stats)
                        k += int(relu)
                    else:
                        input.features = _run(module, input.features)
            else:
                input = _run(module, input)
        if residual is not None:
            input.features = input.features + residual
        return input
