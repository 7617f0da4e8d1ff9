"""Config plumbing for the R3M surface: the reference composes `cfgs/config_rep.yaml` with hydra 1.1 and instantiates
`agent._target_: r3m.R3M` (/root/reference/r3m/cfgs/config_rep.yaml:30-41, train_representation.py:28, __init__.py:69-71).
hydra / omegaconf are optional here: this module is a small PyYAML resolver with the same behaviour for what the path
needs — `key=value` / `a.b=value` overrides, `${key}` interpolation, `_target_` instantiation, attribute access.
"""
import copy
import importlib
import re

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")

# `_target_` strings found in reference configs/checkpoints map onto this package
_TARGET_ALIASES = {"r3m.R3M": "r3m_amd.R3M", "r3m.models.models_r3m.R3M": "r3m_amd.R3M"}


class Cfg(dict):
    """dict with attribute access (enough of OmegaConf's DictConfig for train_representation / load_r3m)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lookup(root, dotted):
    cur = root
    for part in dotted.split("."):
        cur = cur[part]
    return cur


def _resolve(node, root):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve(node[k], root)
        return node
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:  # whole-value interpolation keeps the referenced type (${lr} stays a float)
            return _resolve(copy.deepcopy(_lookup(root, m.group(1).strip())), root)
        return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1).strip()), root)), node)
    return node


def _parse_scalar(text):
    v = yaml.safe_load(text)
    if isinstance(v, str):
        # YAML 1.1 reads "1e-4" as a string; hydra/OmegaConf read it as a float
        try:
            return float(v)
        except ValueError:
            return v
    return v


def _fix_floats(node):
    if isinstance(node, dict):
        return {k: _fix_floats(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_fix_floats(v) for v in node]
    if isinstance(node, str) and re.fullmatch(r"[+-]?\d+(\.\d*)?[eE][+-]?\d+", node):
        return float(node)
    return node


def load_config(path, overrides=()):
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    raw.pop("defaults", None)  # hydra composition list (launcher / output dir): cluster glue, out of scope
    raw = _fix_floats(raw)
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not key=value")
        k, v = ov.split("=", 1)
        cur = raw
        parts = k.lstrip("+").split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = _parse_scalar(v)
    return _wrap(_resolve(raw, raw))


def instantiate(node, **extra):
    """hydra.utils.instantiate for the one shape the path uses: {_target_: pkg.Class, **kwargs}."""
    node = dict(node)
    target = node.pop("_target_")
    target = _TARGET_ALIASES.get(target, target)
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    node.update(extra)
    return cls(**node)
