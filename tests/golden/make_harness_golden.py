"""Generates tests/golden/harness_golden.json in the BUILD container (needs /root/reference).

Golden vectors for the training harness, produced by the REFERENCE's own Python:
  * util/config.py: cfg_from_yaml_file (`_BASE_CONFIG_` composition) on the shipped experiment configs
    and cfg_from_list (`--set`) on a handful of overrides -> parsed dictionaries;
  * util/common_utils.py: step / poly / cos learning-rate functions and adjust_lr on a grid of
    (epoch, iteration) points, and build_optimizer's hyper-parameters.
The reference modules are imported from where they lie; only third-party imports that this image lacks
and that the functions under test never touch (easydict, open3d, SharedArray, PIL, tensorboardX) are
given empty stand-in modules — EasyDict by a 15-line attribute dictionary with its documented behaviour.
Nothing of the reference's source is copied: the JSON holds inputs and outputs only."""
import json
import os
import sys
import types

REF = "/root/reference"


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = type(value)(self.__class__(x) if isinstance(x, dict) else x for x in value)
        elif isinstance(value, dict) and not isinstance(value, EasyDict):
            value = self.__class__(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def update(self, e=None, **f):
        for k, v in dict(e or {}, **f).items():
            setattr(self, k, v)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def main():
    _stub("easydict", EasyDict=EasyDict)
    for name in ("open3d", "SharedArray", "tensorboardX"):
        _stub(name)
    _stub("PIL", Image=types.SimpleNamespace())
    _stub("PIL.Image")
    sys.path.insert(0, REF)
    os.chdir(REF)   # the reference opens _BASE_CONFIG_ paths relative to the working directory
    from util import config as rcfg
    from util import common_utils as rcu
    import torch

    def plain(v):
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v

    out = {"configs": {}, "set": [], "lr": [], "optim": []}
    cfg_files = ["cfgs/scannet/spconv.yaml", "cfgs/da_front3d_scannet/spconv.yaml",
                 "cfgs/da_front3d_scannet/spconv_st.yaml", "cfgs/da_front3d_s3dis/spconv_st.yaml"]
    for f in cfg_files:
        c = rcfg.cfg_from_yaml_file(f, EasyDict())
        out["configs"][f] = plain(c)
    sets = [["OPTIMIZATION.base_lr", "0.05", "OPTIMIZATION.lr_decay", "poly"],
            ["MODEL.BACKBONE.mid_channel", "32", "EVALUATION.evaluate", "False"],
            ["DATA_CONFIG.DATA_AUG.aug_list", "scene_aug,crop"],
            ["MODEL.BACKBONE", "block_reps:3,mid_channel:32"]]
    for sl in sets:
        c = rcfg.cfg_from_yaml_file("cfgs/scannet/spconv.yaml", EasyDict())
        rcfg.cfg_from_list(list(sl), c)
        out["set"].append({"args": sl, "result": plain(c)})
    # learning rates
    net = torch.nn.Linear(2, 2)
    for decay, extra in (("step", {"step_epoch": 30, "multiplier": 0.5}), ("poly", {}), ("cos", {})):
        ocfg = EasyDict(dict(base_lr=0.01, lr_decay=decay, momentum=0.9, weight_decay=1e-4, **extra))
        opt = rcu.build_optimizer(ocfg, net)
        for epoch in (0, 1, 29, 30, 31, 99):
            for it in (0, 7, 49):
                rcu.adjust_lr(ocfg, opt, None, 100, 50, epoch, it)
                out["lr"].append({"cfg": plain(ocfg), "total_epochs": 100, "iters_per_epoch": 50, "epoch": epoch,
                                  "iter": it, "lr": opt.param_groups[0]["lr"]})
    for kind in ("sgd", "adam", "adamw"):
        ocfg = EasyDict(dict(optim=kind, base_lr=0.02, momentum=0.8, weight_decay=5e-4))
        opt = rcu.build_optimizer(ocfg, net)
        g = opt.param_groups[0]
        out["optim"].append({"cfg": plain(ocfg), "class": type(opt).__name__, "lr": g["lr"],
                             "momentum": g.get("momentum"), "weight_decay": g.get("weight_decay")})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_golden.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
