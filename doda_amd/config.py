"""YAML experiment configuration with the reference's semantics (util/config.py:21-85).

* `cfg_from_yaml_file(path, cfg)`: load a YAML file into an attribute dictionary; a mapping that holds
  `_BASE_CONFIG_: <yaml path>` first pulls that file in (recursively) and then overrides it with its
  own keys (util/config.py:56-74) — this is how cfgs/scannet/spconv.yaml composes
  cfgs/dataset_cfgs/scannet/scannet_cfg.yaml into DATA_CONFIG / DATA_CONFIG_TAR.  Base paths are resolved
  as the reference does (relative to the working directory) and, failing that, relative to `root`
  (the directory that contains `cfgs/`), so the configs work from any directory.
* `cfg_from_list(["A.b", "v", ...], cfg)`: the `--set` override list (util/config.py:21-53): keys must
  exist, values go through literal_eval, `k1:v1,k2:v2` updates a sub-dictionary, `a,b,c` a list, and
  the type of a scalar must match the type it replaces.

`Config` replaces EasyDict (attribute access on nested dictionaries, nested dicts converted on
assignment); the parsed result equals the reference's for the shipped cfgs — tests/golden/
harness_golden.json holds what the reference's own parser produced."""
import os
from ast import literal_eval

import yaml


class Config(dict):
    """dict with attribute access; nested dicts (also inside lists) become Config on assignment."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, (list, tuple)):
            return type(v)(Config._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kwargs):
        for k, v in dict(other or {}, **kwargs).items():
            self[k] = v

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            return v
        return plain(self)


def _load(path, root=None):
    for cand in ([path] if os.path.isabs(path) else [path] + ([os.path.join(root, path)] if root else [])):
        if os.path.exists(cand):
            with open(cand, "r") as f:
                return yaml.load(f, Loader=yaml.FullLoader)
    raise FileNotFoundError("config file %r not found (cwd %s, root %s)" % (path, os.getcwd(), root))


def merge_new_config(config, new_config, root=None):
    if "_BASE_CONFIG_" in new_config:
        base = _load(new_config["_BASE_CONFIG_"], root)
        config.update(Config(base))
        merge_new_config(config, base, root)
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = Config()
        merge_new_config(config[key], val, root)
    return config


def cfg_from_yaml_file(cfg_file, config=None, root=None):
    """root: directory containing `cfgs/` (defaults to the grandparent of a `cfgs/<group>/<name>.yaml`)."""
    config = Config() if config is None else config
    if root is None:
        parts = os.path.abspath(cfg_file).split(os.sep)
        if "cfgs" in parts:
            root = os.sep.join(parts[:len(parts) - 1 - parts[::-1].index("cfgs")])
    merge_new_config(config, _load(cfg_file, root), root)
    return config


def cfg_from_list(cfg_list, config):
    """`--set KEY VALUE [KEY VALUE ...]` (reference util/config.py:21-53)."""
    assert len(cfg_list) % 2 == 0, "--set takes KEY VALUE pairs"
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split(".")
        d = config
        for subkey in key_list[:-1]:
            assert subkey in d, "NotFoundKey: %s" % subkey
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d, "NotFoundKey: %s" % subkey
        try:
            value = literal_eval(v)
        except (ValueError, SyntaxError):
            value = v
        if type(value) != type(d[subkey]) and isinstance(d[subkey], dict):
            for src in value.split(","):
                cur_key, cur_val = src.split(":")
                d[subkey][cur_key] = type(d[subkey][cur_key])(cur_val)
        elif type(value) != type(d[subkey]) and isinstance(d[subkey], list):
            val_list = value.split(",")
            d[subkey] = [type(d[subkey][0])(x) for x in val_list]
        else:
            assert type(value) == type(d[subkey]), \
                "type {} does not match original type {}".format(type(value), type(d[subkey]))
            d[subkey] = value
    return config
