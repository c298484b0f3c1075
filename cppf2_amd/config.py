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


def load_config(config_dir="config", config_name="config", overrides=(), with_hydra=False):
    """Resolves `defaults:` groups and applies `a.b=value` overrides.  Returns a Cfg; with_hydra=True returns (cfg, hydra
    node) -- the `hydra:` block (run.dir etc., config/config.yaml:16-22) is not part of the cfg the code sees, as with hydra."""
    if not os.path.isabs(config_dir) and not os.path.isdir(config_dir):
        # a relative directory that does not exist under the working directory: the repository's own config/ tree (the entry
        # points are started from anywhere; hydra resolves config_path relative to the script, train_shot.py:159)
        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), config_dir)
        if os.path.isdir(here):
            config_dir = here
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
    hydra = cfg.pop("hydra", None) or {}
    if with_hydra:
        return _wrap(cfg), _wrap(hydra)
    return _wrap(cfg)


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def run_dir(cfg, hydra_node=None, default="checkpoints/${cat_name}"):
    """hydra.run.dir with its ${key} interpolations resolved against the cfg (config/config.yaml:17-18:
    checkpoints/${cat_name}): the directory the reference's trainers write into."""
    import re
    d = default
    if hydra_node and isinstance(hydra_node.get("run"), dict) and hydra_node["run"].get("dir"):
        d = str(hydra_node["run"]["dir"])

    def sub(m):
        node = cfg
        for part in m.group(1).split("."):
            node = node[part]
        return str(node)
    return re.sub(r"\$\{([^}]+)\}", sub, d)


def save_run_config(cfg, directory):
    """<run dir>/.hydra/config.yaml: the resolved cfg, where eval.py:92,97 (omegaconf.OmegaConf.load) looks for it."""
    os.makedirs(os.path.join(directory, ".hydra"), exist_ok=True)
    path = os.path.join(directory, ".hydra", "config.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(_plain(cfg), f, default_flow_style=None, sort_keys=False)
    return path


def load_checkpoint_config(path):
    """A checkpoint run dir's .hydra/config.yaml (eval.py:92,97); legacy keys are kept but ignored by code."""
    with open(path) as f:
        return _wrap(yaml.safe_load(f) or {})
