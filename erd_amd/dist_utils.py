"""The path's two collectives besides the gradient all-reduce (mmdet/utils/dist_utils.py:59-65).  Device-agnostic:
RCCL on the GPUs (backend 'nccl'), gloo in the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def reduce_mean(tensor: torch.Tensor) -> torch.Tensor:
    """`reduce_mean` of the reference: the mean over ranks as divide-by-world then all-reduce(SUM); the identity when no
    process group is initialised (dist_utils.py:61-62 -- which is how every single-process head test runs).  The ERD head
    calls it ONCE per step on the 2-float vector [sum num_pos, sum weight_targets] (the reference: twice, each followed by
    `.item()`, gfl_head_increment_erd.py:390-391,406-407); the result stays on the device."""
    if world_size() == 1:
        return tensor
    out = tensor.clone()
    out.div_(world_size())
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out
