"""yacs-compatible config node on top of PyYAML (yacs is not installed here).

Covers what the reference uses (config.py:5-32 and the presence tests in the
method constructors, e.g. `'stage' in config`, BCNN.py:36): attribute and item
access, `in`, `load_cfg(file_or_str)`, `freeze()` / immutability, `__str__` as
yaml, `clone()`.
"""
import argparse
import copy
import io

import yaml


class CfgNode(dict):
    IMMUTABLE = '__immutable__'

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)

    # attribute access ----------------------------------------------------
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f'Attempted to set {name} to {value}, but CfgNode is immutable')
        self[name] = value

    def __setitem__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f'Attempted to set {name} to {value}, but CfgNode is immutable')
        dict.__setitem__(self, name, CfgNode(value) if isinstance(value, dict) and not isinstance(value, CfgNode) else value)

    # freezing ------------------------------------------------------------
    def is_frozen(self):
        return self.__dict__.get(CfgNode.IMMUTABLE, False)

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})
        out._set_immutable(self.is_frozen())
        return out

    # (de)serialisation -----------------------------------------------------
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(self.to_dict(), **kw)

    def __str__(self):
        return self.dump(default_flow_style=False, sort_keys=False)

    def __repr__(self):
        return f'CfgNode({dict.__repr__(self)})'

    @classmethod
    def load_cfg(cls, cfg_file_obj_or_str):
        if isinstance(cfg_file_obj_or_str, (str, bytes)):
            data = yaml.safe_load(io.StringIO(cfg_file_obj_or_str if isinstance(cfg_file_obj_or_str, str)
                                              else cfg_file_obj_or_str.decode()))
        else:
            data = yaml.safe_load(cfg_file_obj_or_str)
        return cls(data or {})


def load_config(path):
    with open(path) as f:
        cfg = CfgNode.load_cfg(f)
    cfg.freeze()
    return cfg


def setup_config(argv=None):
    """`--config path.yaml` -> frozen CfgNode (reference config.py:5-18; default configs/Baseline.yaml :28-32)."""
    parser = argparse.ArgumentParser(description='Hawkeye (MI355X-native heads)')
    parser.add_argument('--config', default=None, type=str, help='path to config file')
    args, _ = parser.parse_known_args(argv)
    return load_config(args.config if args.config is not None else 'configs/Baseline.yaml')
