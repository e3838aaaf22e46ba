"""Minimal registry with the surface the reference's configs use (mmdet/registry.py:35-121 exposes 17
mmengine registries; the ERD path needs two): ``@MODELS.register_module()`` / ``MODELS.build(cfg)`` where
``cfg`` is a dict whose ``type`` names the class and the remaining keys are constructor kwargs."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._module_dict: Dict[str, type] = {}

    def __contains__(self, key: str) -> bool:
        return key in self._module_dict

    def get(self, key: str) -> Optional[type]:
        return self._module_dict.get(key)

    def register_module(self, name: Optional[str] = None, force: bool = False, module: Optional[type] = None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg: Dict[str, Any], default_args: Optional[Dict[str, Any]] = None, **kwargs) -> Any:
        if cfg is None:
            return None
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg must be a dict, got {type(cfg)}")
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        if "type" not in args:
            raise KeyError(f"`cfg` must contain the key 'type', got {cfg}")
        t = args.pop("type")
        if isinstance(t, str):
            cls = self._module_dict.get(t)
            if cls is None:
                raise KeyError(f"{t} is not in the {self.name} registry (erd_amd builds the ERD hot path only: "
                               f"{sorted(self._module_dict)})")
        else:
            cls = t
        return cls(**args)


MODELS = Registry("model")
TASK_UTILS = Registry("task util")
