"""SGD with the whole parameter update in one native launch.

`FusedSGD` IS `torch.optim.SGD` (same constructor, param groups, `state[p]["momentum_buffer"]`,
`state_dict()` / `load_state_dict()` — the reference's checkpoints store `optimizer.state_dict()`,
util/model_utils.py:87-94, tool/train.py:255-262) with `step()` re-routed: all (parameter, gradient,
momentum buffer) triples of a group go to `doda_sgd_multi` together instead of through torch's
multi-tensor kernels (five launches for the U-Net's 281 tensors).  The arithmetic follows torch's fused
functor term by term; parameters and buffers come out bit-identical (tests/test_gpu_round2.py).

Parameters and gradients must be contiguous fp32 tensors on one HIP device (the native call checks every
tensor on every step and raises otherwise); sparse gradients are refused: use torch.optim.SGD for those."""
import torch

from . import ops as _ops
from ._ext import ext as _ext


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *,
                 maximize=False):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov, maximize=maximize)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._doda_lists = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._doda_lists = {}

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad(set_to_none=True) walks the parameters in Python (0.2 ms per step for the U-Net's 203); one
        extension call does the same stores.  set_to_none=False (in-place zeroing) is torch's."""
        if not set_to_none or _ext is None or not hasattr(_ext, "clear_grads"):
            return super().zero_grad(set_to_none=set_to_none)
        for group in self.param_groups:
            _ext.clear_grads(group["params"])

    def _slow_lists(self, group):
        """(params, grads, buffers, first flags) of the parameters that have a gradient; creates the missing
        momentum buffers (torch: clone of the first gradient — here the kernel writes it, first flag set)."""
        momentum = group["momentum"]
        ps, gs, bs, fs = [], [], [], []
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise RuntimeError("FusedSGD does not take sparse gradients; use torch.optim.SGD")
            first, buf = False, None
            if momentum != 0:
                state = self.state[p]
                buf = state.get("momentum_buffer")
                if buf is None:
                    buf = state["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                    first = True
            ps.append(p); gs.append(p.grad); bs.append(buf); fs.append(int(first))
        return ps, gs, bs, fs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lists = self.__dict__.setdefault("_doda_lists", {})
        for gi, group in enumerate(self.param_groups):
            lr = group["lr"]
            if torch.is_tensor(lr):
                lr = float(lr)
            momentum = group["momentum"]
            params = group["params"]
            # steady state: every parameter has a gradient and a buffer — the lists of the previous step stand
            # (the per-tensor dtype / layout / device checks are made by the native call on every step)
            cached = lists.get(gi)
            grads = [p.grad for p in params]
            if cached is not None and cached[0] is params and len(cached[1]) == len(params) and all(g is not None for g in grads):
                ps, bs, fs = params, cached[1], cached[2]
                gs = grads
            else:
                ps, gs, bs, fs = self._slow_lists(group)
                if len(ps) == len(params) and not any(fs):
                    lists[gi] = (params, bs, fs)
            if not ps:
                continue
            if _ext is not None:
                _ext.sgd_step(ps, gs, bs if momentum != 0 else [], fs, float(lr), float(momentum),
                              float(group["dampening"]), float(group["weight_decay"]), bool(group["nesterov"]),
                              bool(group["maximize"]))
            else:
                _ops.sgd_multi(ps, gs, bs, fs, lr, momentum, group["dampening"], group["weight_decay"],
                               group["nesterov"], group["maximize"])
        return loss
