"""Python-file configs with ``_base_`` inheritance (the mmengine ``Config`` surface the ERD path touches:
``Config.fromfile`` -- also called at model-build time on ``ori_setting.ori_config_file``,
gfl_increment_erd.py:105 -- recursive dict merge, attribute access, ``_delete_``)."""
from __future__ import annotations

import copy
import os
from typing import Any, Dict


class ConfigDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            dict.__setitem__(self, k, _wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'") from e

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, _wrap(v))

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self) -> Dict[str, Any]:
        return _unwrap(self)


def _wrap(v):
    if isinstance(v, ConfigDict):
        return v
    if isinstance(v, dict):
        return ConfigDict(v)
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_unwrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_unwrap(x) for x in v)
    return v


def _merge(base: dict, child: dict) -> dict:
    """child overrides base, dicts merge recursively, `_delete_=True` replaces."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = v
    return out


def _load_py(path: str) -> dict:
    ns: Dict[str, Any] = {"__file__": path}
    with open(path, "r") as f:
        code = compile(f.read(), path, "exec")
    exec(code, ns)
    import types
    return {k: v for k, v in ns.items()
            if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}


def _load_with_bases(path: str) -> dict:
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    body = _load_py(path)
    bases = body.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged: dict = {}
    for b in bases:
        bcfg = _load_with_bases(os.path.join(os.path.dirname(path), b))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f"Duplicate key is not allowed among bases: {sorted(dup)}")
        merged.update(bcfg)
    return _merge(merged, body)


class Config(ConfigDict):
    @staticmethod
    def fromfile(filename: str) -> "Config":
        cfg = Config(_load_with_bases(str(filename)))
        dict.__setitem__(cfg, "filename", os.path.abspath(str(filename)))
        return cfg

    def merge_from_dict(self, options: Dict[str, Any]) -> None:
        """`--cfg-options a.b.c=v` style overrides (tools/train.py:72-73)."""
        for key, v in options.items():
            d = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in d:
                    d[p] = ConfigDict()
                d = d[p]
            d[parts[-1]] = v
