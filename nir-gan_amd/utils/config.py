"""YAML -> attribute tree, standing in for OmegaConf.load (train.py:32-42) which is not required here.

Supports what the reference does with its config object: attribute access at any depth,
``"key" in cfg``, ``dict(cfg.section)`` and item access.
"""
import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_attr(v) for v in x]
    return x


def load_config(path: str) -> AttrDict:
    with open(path) as f:
        return to_attr(yaml.safe_load(f))
