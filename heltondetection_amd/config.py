"""Config files (reference README.md:12 "all hyper-parameters of model definition, training, evaluation and
testing are selected by config files").  The reference's own format is not in the snapshot, so this
is this repo's contract, shaped after that sentence:

    a config is one mapping with the sections  model / data / train / eval / test / log;
    `model` is a registry config (dict with `type`, nested dicts with `type` are built recursively by their
    parent — registry.py); the other sections are plain keyword sets read by runners.py.

Formats: .yaml / .yml (PyYAML), .json, or .py (module-level names that do not start with "_" become the mapping —
the style the detection frameworks the README credits use).  `_base_: other.yaml` (or a list) inherits, nearest
file wins, dicts merge recursively, a value of `_delete_: true` inside a dict drops the inherited keys first.
Command-line overrides are dotted paths: `train.lr=0.002`, `model.head.num_classes=20`, parsed as YAML scalars.
"""
from __future__ import annotations

import copy
import json
import os
import runpy
from typing import Any, Dict, Iterable, List

SECTIONS = ("model", "data", "train", "eval", "test", "log")

DEFAULTS: Dict[str, Dict[str, Any]] = {
    "data": {"type": "synthetic_coco", "root": None, "n_images": 64, "val_images": 16, "size": 640, "num_classes": 80,
             "style": "coco", "max_boxes": 40, "paint": False, "seed": 0, "batch": 16},
    "train": {"epochs": 1, "max_steps": None, "lr": 1e-3, "weight_decay": 5e-4, "warmup_steps": 10, "amp": "bf16",
              "hip_graph": True, "mosaic": 0.0, "bucket_mb": 0, "log_every": 10, "eval_every": 0, "save_every": 1,
              "seed": 0, "miopen_find": False},
    "eval": {"conf_thr": 0.001, "iou_thr": 0.6, "max_det": 300, "max_nms": 30000, "batch": None, "amp": "bf16"},
    "test": {"conf_thr": 0.25, "iou_thr": 0.45, "max_det": 300, "max_nms": 30000, "batch": None, "amp": "bf16"},
    "log": {"jsonl": True, "stdout": True, "tensorboard": "auto"},
}


class ConfigError(ValueError):
    pass


def _merge(base: Dict[str, Any], over: Dict[str, Any]) -> Dict[str, Any]:
    out = copy.deepcopy(base)
    if over.get("_delete_"):
        out = {}
    for k, v in over.items():
        if k == "_delete_":
            continue
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _read(path: str) -> Dict[str, Any]:
    ext = os.path.splitext(path)[1].lower()
    if ext in (".yaml", ".yml"):
        import yaml
        with open(path) as f:
            d = yaml.safe_load(f) or {}
    elif ext == ".json":
        with open(path) as f:
            d = json.load(f)
    elif ext == ".py":
        ns = runpy.run_path(path)
        d = {k: v for k, v in ns.items()
             if not k.startswith("_") and isinstance(v, (dict, list, tuple, str, int, float, bool, type(None)))}
        if "_base_" in ns:
            d["_base_"] = ns["_base_"]
    else:
        raise ConfigError(f"{path}: unknown config format {ext!r} (use .yaml, .json or .py)")
    if not isinstance(d, dict):
        raise ConfigError(f"{path}: the top level of a config must be a mapping")
    return d


def _load_with_bases(path: str, seen: tuple = ()) -> Dict[str, Any]:
    path = os.path.abspath(path)
    if path in seen:
        raise ConfigError(f"_base_ cycle through {path}")
    if not os.path.exists(path):
        raise ConfigError(f"config file not found: {path}")
    d = _read(path)
    bases = d.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    out: Dict[str, Any] = {}
    for b in bases:
        out = _merge(out, _load_with_bases(os.path.join(os.path.dirname(path), b), seen + (path,)))
    return _merge(out, d)


def apply_overrides(cfg: Dict[str, Any], overrides: Iterable[str]) -> Dict[str, Any]:
    import yaml
    cfg = copy.deepcopy(cfg)
    for item in overrides or ():
        if "=" not in item:
            raise ConfigError(f"override {item!r} is not of the form section.key=value")
        key, val = item.split("=", 1)
        node = cfg
        parts = key.strip().split(".")
        for p in parts[:-1]:
            nxt = node.get(p)
            if not isinstance(nxt, dict):
                nxt = node[p] = {}
            node = nxt
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def load_config(path: str, overrides: Iterable[str] = ()) -> Dict[str, Any]:
    """File (+ its _base_ chain) -> defaults filled in -> overrides applied -> validated."""
    cfg = _load_with_bases(path)
    unknown = [k for k in cfg if k not in SECTIONS]
    if unknown:
        raise ConfigError(f"{path}: unknown top-level section(s) {unknown}; known: {list(SECTIONS)}")
    for sec, dflt in DEFAULTS.items():
        cfg[sec] = _merge(dflt, cfg.get(sec) or {})
    cfg = apply_overrides(cfg, overrides)
    validate(cfg, path)
    return cfg


def validate(cfg: Dict[str, Any], where: str = "config") -> None:
    m = cfg.get("model")
    if not isinstance(m, dict) or "type" not in m:
        raise ConfigError(f"{where}: section `model` must be a registry config with a `type` key")
    from .registry import DETECTORS
    from . import detectors  # noqa: F401  (registers the classes)
    if m["type"] not in DETECTORS:
        raise ConfigError(f"{where}: model.type {m['type']!r} is not a registered detector; known: {DETECTORS.names()}")
    for sec, dflt in DEFAULTS.items():
        extra = [k for k in cfg[sec] if k not in dflt]
        if extra and sec != "data":  # data sets may carry reader-specific keys
            raise ConfigError(f"{where}: unknown key(s) {extra} in section `{sec}`; known: {sorted(dflt)}")
    if cfg["train"]["amp"] not in ("bf16", "fp32", None):
        raise ConfigError(f"{where}: train.amp must be bf16 or fp32")
    if cfg["data"]["type"] not in ("synthetic_coco", "coco"):
        raise ConfigError(f"{where}: data.type must be synthetic_coco or coco")


def dump(cfg: Dict[str, Any]) -> str:
    import yaml
    return yaml.safe_dump(cfg, sort_keys=False)


def sections(cfg: Dict[str, Any]) -> List[str]:
    return [s for s in SECTIONS if s in cfg]
