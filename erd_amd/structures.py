"""Containers at the path's boundary: each data sample exposes ``.gt_instances.bboxes
[G,4] xyxy px``, ``.gt_instances.labels [G] int64``, optional ``.ignored_instances`` and ``.metainfo``
with ``img_shape`` / ``pad_shape`` / ``batch_input_shape`` (models/utils/misc.py:89-122)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple


class InstanceData:
    def __init__(self, metainfo: Optional[Dict[str, Any]] = None, **kwargs):
        object.__setattr__(self, "_data", {})
        object.__setattr__(self, "_meta", dict(metainfo or {}))
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self._data[k] = v

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_data")
        if k in d:
            return d[k]
        m = object.__getattribute__(self, "_meta")
        if k in m:
            return m[k]
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self._data or k in self._meta

    def __len__(self):
        for v in self._data.values():
            return len(v)
        return 0

    def keys(self):
        return list(self._data.keys())

    def __getitem__(self, item) -> "InstanceData":
        """index every field alike (mask / index tensor / slice), as mmengine's InstanceData does"""
        out = InstanceData(metainfo=self._meta)
        for k, v in self._data.items():
            setattr(out, k, v[item])
        return out

    @property
    def metainfo(self):
        return dict(self._meta)


class DetDataSample:
    def __init__(self, metainfo: Optional[Dict[str, Any]] = None):
        self._meta = dict(metainfo or {})
        self.gt_instances: Optional[InstanceData] = None
        self.ignored_instances: Optional[InstanceData] = None
        self.pred_instances: Optional[InstanceData] = None

    @property
    def metainfo(self):
        return dict(self._meta)

    def set_metainfo(self, meta: Dict[str, Any]) -> None:
        self._meta.update(meta)

    def __contains__(self, k):
        return getattr(self, k, None) is not None


def unpack_gt_instances(batch_data_samples) -> Tuple[List[InstanceData], List[Optional[InstanceData]], List[dict]]:
    """models/utils/misc.py:89-122."""
    gts, ignored, metas = [], [], []
    for ds in batch_data_samples:
        metas.append(ds.metainfo)
        gts.append(ds.gt_instances)
        ignored.append(ds.ignored_instances if "ignored_instances" in ds else None)
    return gts, ignored, metas
