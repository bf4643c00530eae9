"""The slice of Hydra/OmegaConf the reference's config surface needs (hydra-core / omegaconf are not
installed): YAML loading, ``${a.b.c}`` interpolation, ``key=value`` CLI overrides and recursive
``_target_`` instantiation (reference experiments/train.py:8-25, conf/*.yaml, edm.py:169)."""
from __future__ import annotations

import importlib
import os
import re
from typing import Any, List, Optional

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


class Config(dict):
    """dict with attribute access (cfg.model.lr) like DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Config({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lookup(root, path: str):
    cur = root
    for part in path.split("."):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


def _resolve(node, root):
    if isinstance(node, dict):
        return Config({k: _resolve(v, root) for k, v in node.items()})
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:
            return _resolve(_lookup(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root)), node)
    return node


def _set(cfg, dotted: str, value):
    parts = dotted.lstrip("+").split(".")
    cur = cfg
    for p in parts[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            cur[p] = Config()
        cur = cur[p]
    cur[parts[-1]] = value


def compose(config_name: str, config_path: str, overrides: Optional[List[str]] = None) -> Config:
    """Load ``<config_path>/<config_name>.yaml``, apply ``a.b=c`` overrides (YAML-typed), resolve ``${}``."""
    fn = os.path.join(config_path, config_name if config_name.endswith(".yaml") else config_name + ".yaml")
    with open(fn) as f:
        raw = _wrap(yaml.safe_load(f))
    for ov in overrides or []:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not key=value")
        k, v = ov.split("=", 1)
        _set(raw, k, _wrap(yaml.safe_load(v)))
    return _resolve(raw, raw)


def to_container(cfg) -> Any:
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [to_container(v) for v in cfg]
    return cfg


# `_target_` paths of the reference's YAMLs that name classes of packages absent here (lightning): resolved to the
# stand-ins of this build, so conf/*.yaml of the reference instantiate unchanged
_ALIASES = {
    "lightning.pytorch.callbacks.ModelCheckpoint": "tinyedm_amd.callbacks.ModelCheckpoint",
    "lightning.pytorch.callbacks.model_checkpoint.ModelCheckpoint": "tinyedm_amd.callbacks.ModelCheckpoint",
    "lightning.Trainer": "tinyedm_amd.trainer.Trainer",
    "lightning.pytorch.Trainer": "tinyedm_amd.trainer.Trainer",
}


def _locate(target: str):
    target = _ALIASES.get(target, target)
    mod, _, name = target.rpartition(".")
    parts = target.split(".")
    for i in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        for attr in parts[i:]:
            obj = getattr(obj, attr)
        return obj
    raise ImportError(f"cannot locate {target!r}")


def instantiate(cfg, **overrides):
    """Recursive ``_target_`` instantiation (hydra.utils.instantiate semantics for dict configs)."""
    if isinstance(cfg, list):
        return [instantiate(v) for v in cfg]
    if not isinstance(cfg, dict):
        return cfg
    if "_target_" not in cfg:
        return Config({k: instantiate(v) for k, v in cfg.items()})
    kwargs = {k: instantiate(v) for k, v in cfg.items() if k not in ("_target_", "_partial_", "_recursive_")}
    kwargs.update(overrides)
    return _locate(cfg["_target_"])(**kwargs)
