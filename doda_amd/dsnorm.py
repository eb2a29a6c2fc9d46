"""Domain-specific BatchNorm (reference model/dsnorm.py:12-315, DSNorm / DSNorm1d, convert_dsnorm,
set_ds_source / set_ds_target, BatchNorm-checkpoint loading): one affine pair
shared by both domains, separate running statistics for the source (domain_label 0) and the target
(domain_label 1) domain.  State-dict keys are the reference's (`weight`, `bias`,
`running_mean_source|target`, `running_var_source|target`, `num_batches_tracked`).

On device [M, C] features the forward goes through the same fused HIP BatchNorm(+ReLU) kernels as
nn.BatchNorm1d: doda_amd.spconv.SparseSequential recognises this class AND the reference's own class
(by its attributes), so `cfgs/*` that select DSNorm run on the fused path too."""
import torch
from torch import nn
from torch.nn import functional as F


class DSNorm1d(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.domain_label = 0   # 0: source, 1: target
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        if track_running_stats:
            for dom in ("source", "target"):
                self.register_buffer("running_mean_" + dom, torch.zeros(num_features))
                self.register_buffer("running_var_" + dom, torch.ones(num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            for name in ("running_mean_source", "running_mean_target", "running_var_source",
                         "running_var_target", "num_batches_tracked"):
                self.register_parameter(name, None)

    def set_domain_label(self, domain_label):
        self.domain_label = domain_label

    def reset_running_stats(self):
        if self.track_running_stats:
            self.running_mean_source.zero_()
            self.running_var_source.fill_(1)
            self.running_mean_target.zero_()
            self.running_var_target.fill_(1)
            self.num_batches_tracked.zero_()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        """Reference model/dsnorm.py:87-170: a key `running_*_source|target` that the checkpoint lacks is
        read from the plain BatchNorm key (`running_mean` / `running_var`), so a source-only
        nn.BatchNorm1d checkpoint initialises BOTH domains; a missing `num_batches_tracked` counts as 0.
        Keys the module does not know are not reported as unexpected (the reference's check is
        commented out, :164-170)."""
        if self.track_running_stats and prefix + "num_batches_tracked" not in state_dict:
            state_dict[prefix + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        local = {k: v for k, v in list(self._parameters.items()) + list(self._buffers.items()) if v is not None}
        with torch.no_grad():
            for name, param in local.items():
                key = prefix + name
                if name.endswith(("_source", "_target")) and key not in state_dict:
                    key = key[:-7]
                if key not in state_dict:
                    if strict:
                        missing_keys.append(key)
                    continue
                value = state_dict[key]
                if param.dim() == 0 and value.dim() == 1:
                    value = value[0]
                if value.shape != param.shape:
                    error_msgs.append("size mismatch for {}: copying a param with shape {} from checkpoint, "
                                      "the shape in current model is {}.".format(key, value.shape, param.shape))
                    continue
                param.copy_(value)

    @classmethod
    def convert_dsnorm(cls, module):
        """Reference model/dsnorm.py:172-210 (called at tool/train.py:332, tool/st.py:476,
        tool/test.py:290): every torch BatchNorm layer of `module` becomes a DSNorm carrying the same
        affine parameters; as in the reference both domains start out SHARING the BatchNorm's
        running-statistics tensors (and its batch counter)."""
        out = module
        if isinstance(module, nn.modules.batchnorm._BatchNorm):
            out = cls(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats)
            if module.affine:
                out.weight.data = module.weight.data.clone().detach()
                out.bias.data = module.bias.data.clone().detach()
                out.weight.requires_grad = module.weight.requires_grad
                out.bias.requires_grad = module.bias.requires_grad
            out.running_mean_target = out.running_mean_source = module.running_mean
            out.running_var_target = out.running_var_source = module.running_var
            out.num_batches_tracked = module.num_batches_tracked
        for name, child in module.named_children():
            out.add_module(name, cls.convert_dsnorm(child))
        return out

    def running_stats(self):
        dom = "target" if self.domain_label else "source"
        return getattr(self, "running_mean_" + dom), getattr(self, "running_var_" + dom)

    def forward(self, input):
        if input.dim() not in (2, 3):
            raise ValueError("expected 2D or 3D input (got {}D input)".format(input.dim()))
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked += 1
            if self.momentum is None:
                factor = 1.0 / float(self.num_batches_tracked)
        mean, var = self.running_stats() if self.track_running_stats else (None, None)
        return F.batch_norm(input, mean, var, self.weight, self.bias,
                            self.training or not self.track_running_stats, factor, self.eps)

    def extra_repr(self):
        return "{}, eps={}, momentum={}, affine={}, track_running_stats={}".format(
            self.num_features, self.eps, self.momentum, self.affine, self.track_running_stats)


DSNorm = DSNorm1d


def set_ds_source(m):
    """`model.apply(set_ds_source)`: reference model/dsnorm.py:306-309."""
    if m.__class__.__name__.find("DSNorm") != -1:
        m.set_domain_label(0)


def set_ds_target(m):
    """`model.apply(set_ds_target)`: reference model/dsnorm.py:312-315."""
    if m.__class__.__name__.find("DSNorm") != -1:
        m.set_domain_label(1)
