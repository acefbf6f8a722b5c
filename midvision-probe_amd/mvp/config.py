"""Minimal stand-in for the slice of hydra the trainers use (hydra / omegaconf are not installed):
compose ``configs/<name>.yaml`` with its ``defaults`` list (group choices are ``configs/<group>/<choice>.yaml``, the
reference's hydra layout, so real hydra composes the same tree: ``backbone=dino_b16``), apply ``group=choice``, ``key.sub=value`` and
``+key.sub=value`` overrides (README.md:82 style), and ``instantiate`` a ``_target_`` dict
(configs/backbone/dino_b16.yaml:1, train_depth.py:564-567)."""
from __future__ import annotations

import importlib
import os
from typing import Any, Dict, List

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def _load(path: str) -> dict:
    with open(path) as f:
        return yaml.load(f, Loader=yaml.SafeLoader) or {}


def _set(cfg: dict, dotted: str, value: Any) -> None:
    keys = dotted.split(".")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
    cfg[keys[-1]] = value


def compose(config_name: str, overrides: List[str] = (), config_dir: str = CONFIG_DIR) -> Dict[str, Any]:
    root = _load(os.path.join(config_dir, config_name + ".yaml"))
    groups = {}
    for item in root.pop("defaults", []):
        if isinstance(item, dict):
            groups.update(item)
    plain = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if key.lstrip("+") in groups and "." not in key:
            groups[key.lstrip("+")] = val
        else:
            plain.append((key.lstrip("+"), yaml.load(val, Loader=yaml.SafeLoader)))
    cfg = {}
    for group, choice in groups.items():
        per_file = os.path.join(config_dir, group, f"{choice}.yaml")
        if not os.path.exists(per_file):
            have = sorted(f[:-5] for f in os.listdir(os.path.join(config_dir, group)) if f.endswith(".yaml"))
            raise KeyError(f"{group}={choice}: no {per_file} (choices: {have})")
        cfg[group] = _load(per_file)
    cfg.update(root)
    for key, val in plain:
        _set(cfg, key, val)
    return cfg


def instantiate(node: Dict[str, Any], **kwargs):
    """hydra.utils.instantiate for a flat ``_target_`` mapping."""
    node = dict(node)
    target = node.pop("_target_")
    mod, _, name = target.rpartition(".")
    node.update(kwargs)  # call-site kwargs override config values, as hydra does
    return getattr(importlib.import_module(mod), name)(**node)
