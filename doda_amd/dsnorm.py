"""Domain-specific BatchNorm (reference model/dsnorm.py:12-214, DSNorm / DSNorm1d): one affine pair
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
