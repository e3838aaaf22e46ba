"""The path's collectives (mmdet/utils/dist_utils.py:59-65 and the DDP gradient mean).  Device-agnostic: RCCL on the GPUs
(backend 'nccl'), gloo in the CPU tests -- and gloo with GPU tensors as a CORRECTNESS vehicle: RCCL refuses two ranks on
one device, gloo does not, so `ERD_DIST_BACKEND=gloo` lets the real trainer run at world size 2 on a single MI355X (both
ranks on cuda:0, tests/test_gpu_dist_world2.py).  In that combination a collective is staged through host memory (this
build's gloo is not relied upon to take device tensors); it blocks the host and is not a performance path."""
from __future__ import annotations

import os
from typing import Sequence

import torch
import torch.distributed as dist

BACKEND_ENV = "ERD_DIST_BACKEND"


def backend_name() -> str:
    """the process-group backend the launchers create: 'nccl' (= RCCL on ROCm; configs/_base_/default_runtime.py:14) unless
    ERD_DIST_BACKEND says 'gloo'"""
    b = os.environ.get(BACKEND_ENV, "nccl").lower()
    if b not in ("nccl", "gloo"):
        raise ValueError(f"{BACKEND_ENV}={b!r}: expected 'nccl' or 'gloo'")
    return b


def device_index(local_rank: int) -> int:
    """the GPU of a rank: its own (LOCAL_RANK) under RCCL; with the gloo vehicle ranks may outnumber the GPUs and share them"""
    if backend_name() == "gloo":
        return local_rank % max(torch.cuda.device_count(), 1)
    return local_rank


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def host_staged(tensor: torch.Tensor) -> bool:
    return tensor.is_cuda and dist.get_backend() == "gloo"


class _Done:
    def wait(self) -> None:
        pass


def all_reduce_sum_(tensor: torch.Tensor, async_op: bool = False):
    """in-place SUM over the ranks; with async_op a handle with .wait().  gloo + GPU tensor: device -> host (waits for the
    current stream, which the caller has ordered behind the tensor's producers), all-reduce on the host, host -> device."""
    if host_staged(tensor):
        h = tensor.detach().to("cpu")
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        tensor.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=async_op)


def reduce_mean(tensor: torch.Tensor) -> torch.Tensor:
    """`reduce_mean` of the reference: the mean over ranks as divide-by-world then all-reduce(SUM); the identity when no
    process group is initialised (dist_utils.py:61-62 -- which is how every single-process head test runs).  The ERD head
    calls it ONCE per step on the 2-float vector [sum num_pos, sum weight_targets] (the reference: twice, each followed by
    `.item()`, gfl_head_increment_erd.py:390-391,406-407); the result stays on the device."""
    if world_size() == 1:
        return tensor
    out = tensor.clone()
    out.div_(world_size())
    all_reduce_sum_(out)
    return out


def agree_on_fastest(local_seconds: Sequence[float], group=None, margin: float = 0.015) -> int:
    """Every rank timed the same candidates (ERDTrainer.tune_cu_reserve: warm-up steps at each CU reserve); all ranks must adopt the SAME
    one or their grids differ for the rest of the run.  A step ends when the slowest rank ends, so a candidate is worth the MAX over the
    ranks of its time; a later candidate replaces an earlier one only when it is faster by more than `margin` (1.5 %: three-step
    timings are that noisy, and the first candidate -- reserve 0, the configuration every N = 1 number was taken with -- is the default),
    ties keep the earlier one.  Identical on all ranks by construction (one all-reduce, then local arithmetic on identical numbers).
    Without a process group: the same rule on the local times."""
    t = torch.tensor([float(v) for v in local_seconds], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        t = t.cpu()
    best = 0
    for i in range(1, t.numel()):
        if float(t[i]) < (1.0 - margin) * float(t[best]):
            best = i
    return best
