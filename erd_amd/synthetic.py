"""Synthetic weights and inputs for benchmarking and dry runs (there is no network for checkpoints or COCO).

* `procedural_state_dict(shapes)`: a checkpoint-shaped dict whose every tensor is a pure function of
  (seed, key name, shape) -- numpy PCG64 seeded from sha256("seed:name") -> float32 -- scaled so that activations stay
  O(1) through ~100 layers (He-normal convolutions, BN/GN gains around 1, running_var in [0.5, 1.5], the damped bn3
  gain of the residual branches, `gfl_cls.bias = -log(99)` as gfl_head_increment_erd.py:109-117 initialises it).
  `shapes` comes from the model itself (`{k: v.shape for k, v in model.state_dict().items()}`), so any depth / class
  count works.  tests/ checks that this spec and the oracle's own generator produce identical tensors.
* `demo_batch(...)`: the reference's `demo_mm_inputs` recipe (mmdet/testing/_utils.py:66-75,89-202): RandomState(seed),
  uint8 pixels, 1..9 boxes per image, labels in [0, C_new).
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Mapping, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


def _name_seed(seed: int, name: str) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")


def procedural_tensor(seed: int, name: str, shape: Sequence[int]) -> Tensor:
    shape = tuple(int(s) for s in shape)
    rng = np.random.Generator(np.random.PCG64(_name_seed(seed, name)))
    leaf = name.rsplit(".", 1)[-1]
    f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "project":                                   # Integral's buffer: linspace(0, reg_max, reg_max + 1)
        return torch.linspace(0, shape[0] - 1, shape[0])
    if leaf == "scale":
        return torch.tensor(float(0.9 + 0.2 * rng.random()), dtype=torch.float32)
    if leaf == "running_mean":
        return f32(0.1 * rng.standard_normal(shape))
    if leaf == "running_var":
        return f32(0.5 + rng.random(shape))
    is_norm = (".bn" in name or ".gn." in name or "downsample.1" in name or name.startswith("backbone.bn1")
               or ".bn1." in name)
    if leaf == "weight" and len(shape) == 1:
        return f32((0.25 if ".bn3." in name else 0.75) + 0.5 * rng.random(shape))
    if leaf == "bias" and is_norm:
        return f32(0.1 * rng.standard_normal(shape))
    if leaf == "bias":
        if "gfl_cls" in name:
            return torch.full(shape, -4.59511985013459, dtype=torch.float32)
        return f32(0.05 * rng.standard_normal(shape))
    if leaf == "weight" and len(shape) == 4:
        std = math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        std *= 0.5 if "gfl_cls" in name else (1.5 if "gfl_reg" in name else 1.0)
        return f32(std * rng.standard_normal(shape))
    raise KeyError(f"no synthetic rule for {name} {shape}")


def procedural_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, Tensor]:
    return {k: procedural_tensor(seed, k, shp) for k, shp in shapes.items()}


def state_shapes(model: torch.nn.Module, skip_prefix: str = "ori_model.") -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith(skip_prefix)}


def rand_bboxes(rng: np.random.RandomState, num_boxes: int, w: int, h: int) -> np.ndarray:
    """mmdet/testing/_utils.py:66-75: centre / size drawn uniformly, corners clipped to the image"""
    cx, cy, bw, bh = rng.rand(num_boxes, 4).T
    return np.vstack([(cx * w - w * bw / 2).clip(0, w), (cy * h - h * bh / 2).clip(0, h),
                      (cx * w + w * bw / 2).clip(0, w), (cy * h + h * bh / 2).clip(0, h)]).T


def demo_batch(n: int, h: int = 800, w: int = 1333, num_new_classes: int = 40, seed: int = 0):
    """(uint8 images [3,h,w], xyxy boxes float32, labels int64) x n"""
    rng = np.random.RandomState(seed)
    images, boxes, labels = [], [], []
    for _ in range(n):
        images.append(torch.from_numpy(rng.randint(0, 255, size=(3, h, w), dtype=np.uint8)))
        nb = rng.randint(1, 10)
        boxes.append(torch.from_numpy(rand_bboxes(rng, nb, w, h).astype(np.float32)))
        labels.append(torch.from_numpy(rng.randint(0, num_new_classes, size=nb).astype(np.int64)))
    return images, boxes, labels
