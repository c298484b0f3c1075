"""YAML config surface of the reference (config/config.yaml, config/category/*.yaml, config/custom.yaml) without
hydra/omegaconf: PyYAML + the two hydra features the reference relies on -- a `defaults:` group list and
`key=value` command-line overrides (e.g. `python train_shot.py category=bottle opt.lr=5e-4`)."""
from __future__ import annotations

import os

import yaml


class Cfg(dict):
    """dict with attribute access (cfg.res, cfg.opt.lr), like an OmegaConf node."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, str):
        # YAML 1.1 reads "2e-3" as a string; hydra/omegaconf give a float
        try:
            return float(x) if any(c in x for c in ".eE") and x.replace(".", "").replace("e", "").replace("E", "").replace("-", "").replace("+", "").isdigit() else x
        except ValueError:
            return x
    return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(config_dir="config", config_name="config", overrides=()):
    """Resolves `defaults:` groups and applies `a.b=value` overrides.  Returns a Cfg."""
    with open(os.path.join(config_dir, config_name + ".yaml")) as f:
        raw = yaml.safe_load(f) or {}
    groups = {}
    for d in raw.pop("defaults", []) or []:
        if isinstance(d, dict):
            groups.update(d)
    plain = []
    for o in overrides:
        k, _, v = o.partition("=")
        if k in groups or os.path.isdir(os.path.join(config_dir, k)):
            groups[k] = v
        else:
            plain.append((k, v))
    cfg = dict(raw)
    for g, name in groups.items():
        path = os.path.join(config_dir, g, str(name) + ".yaml")
        with open(path) as f:
            _merge(cfg, yaml.safe_load(f) or {})       # "# @package _global_": merged at the root
    for k, v in plain:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(v)
    cfg.pop("hydra", None)
    return _wrap(cfg)


def load_checkpoint_config(path):
    """A checkpoint run dir's .hydra/config.yaml (eval.py:92,97); legacy keys are kept but ignored by code."""
    with open(path) as f:
        return _wrap(yaml.safe_load(f) or {})
