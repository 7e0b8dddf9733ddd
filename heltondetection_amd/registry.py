"""String-name -> class registries: the plugin surface for backbones, necks, heads, assigners
and detectors (reference README.md:8-9 "swappable / user-definable backbone, neck, head",
README.md:12 "selected by config files").  The reference's own registry is not in the snapshot
(SURVEY.md §8b), so this is this repo's contract: a config is a dict with a ``type`` key naming a
registered class; remaining keys are constructor arguments; nested dicts with ``type`` are built
recursively by the parent."""
from __future__ import annotations

from typing import Any, Callable, Dict


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._items: Dict[str, Callable[..., Any]] = {}

    def register(self, name: str | None = None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._items:
                raise KeyError(f"{key!r} already registered in {self.name}")
            self._items[key] = cls
            return cls
        return deco

    def get(self, key: str):
        if key not in self._items:
            raise KeyError(f"{key!r} is not a registered {self.name}; known: {sorted(self._items)}")
        return self._items[key]

    def build(self, cfg: Dict[str, Any], **extra):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        return self.get(typ)(**cfg, **extra)

    def __contains__(self, key):
        return key in self._items

    def names(self):
        return sorted(self._items)


BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
ASSIGNERS = Registry("assigner")
DETECTORS = Registry("detector")
