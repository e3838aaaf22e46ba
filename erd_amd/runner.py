"""Thin training launcher around ERDTrainer: what ``tools/train.py`` -> mmengine ``Runner``
does for this path and nothing more -- config -> model, epoch loop, LinearLR warm-up + MultiStepLR, 50-iteration
logging, per-epoch checkpoints in the reference's file format, resume.

Reference: tools/train.py:15-129, configs/_base_/schedules/schedule_1x.py:2-28, configs/_base_/default_runtime.py:3-24,
detectors/gfl_increment_erd.py:67-122 (checkpoint keys).
"""
from __future__ import annotations

import bisect
import json
import os
import math
import time
from collections import OrderedDict, deque
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .config import Config
from .engine import ERDTrainer
from .registry import MODELS
from .structures import DetDataSample, InstanceData


# ---------------------------------------------------------------------------------------------------------
# checkpoints: {'meta': ..., 'state_dict': OIHW fp32 tensors keyed as the reference's, 'optimizer': torch-SGD layout}
# ---------------------------------------------------------------------------------------------------------
def model_state_dict(model: torch.nn.Module, with_teacher: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """CPU copy of the state dict in the reference's layout (dense OIHW).  ``with_teacher=False`` drops the frozen
    ``ori_model.*`` copy (it is re-read from ``ori_setting.ori_checkpoint_file`` at build time anyway)."""
    out = OrderedDict()
    for k, v in model.state_dict().items():
        if not with_teacher and k.startswith("ori_model."):
            continue
        out[k] = v.detach().to("cpu").contiguous(memory_format=torch.contiguous_format).clone()
    return out


def save_checkpoint(path: str, model: torch.nn.Module, trainer: Optional[ERDTrainer] = None, meta: Optional[dict] = None,
                    with_teacher: bool = True) -> None:
    if trainer is not None:
        trainer.flush()        # the SGD update of the last step is deferred behind the next teacher forward: apply it
                               # BEFORE the weights are read, or the file pairs stale weights with newer momentum
    ckpt = dict(meta=dict(meta or {}), state_dict=model_state_dict(model, with_teacher))
    if trainer is not None:
        ckpt["optimizer"] = trainer.optimizer_state_dict()
        ckpt["meta"].update(iter=trainer.iter)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)


def load_checkpoint(path: str, model: torch.nn.Module, trainer: Optional[ERDTrainer] = None, strict: bool = True) -> dict:
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    if sd and next(iter(sd)).startswith("module."):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    own = model.state_dict()
    if not any(k.startswith("ori_model.") for k in sd):      # saved without the teacher copy: keep ours
        sd = OrderedDict(sd)
        for k, v in own.items():
            if k.startswith("ori_model."):
                sd[k] = v
    model.load_state_dict(sd, strict=strict)
    if trainer is not None and "optimizer" in ckpt:
        trainer.load_optimizer_state_dict(ckpt["optimizer"])
        trainer.iter = int(ckpt.get("meta", {}).get("iter", 0))
    return ckpt.get("meta", {})


# ---------------------------------------------------------------------------------------------------------
# schedule (schedule_1x.py:7-17)
# ---------------------------------------------------------------------------------------------------------
class ParamSchedule:
    """LinearLR (by iteration) x MultiStepLR (by epoch), composed multiplicatively as mmengine's scheduler list."""

    def __init__(self, param_scheduler: Iterable[dict]):
        self.linear, self.multistep = None, None
        for s in param_scheduler or []:
            s = dict(s)
            t = s.pop("type")
            if t == "LinearLR":
                if s.get("by_epoch", True):
                    raise NotImplementedError("LinearLR by_epoch=True is not used by the ERD configs")
                self.linear = (float(s.get("start_factor", 1 / 3)), float(s.get("end_factor", 1.0)),
                               int(s.get("begin", 0)), int(s["end"]))
            elif t == "MultiStepLR":
                self.multistep = (sorted(int(m) for m in s["milestones"]), float(s.get("gamma", 0.1)))
            else:
                raise NotImplementedError(f"param scheduler {t}")

    def iter_factor(self, it: int) -> float:
        if self.linear is None:
            return 1.0
        s0, s1, b, e = self.linear
        if it >= e:
            return s1
        # mmengine/torch LinearLR: factor interpolates over (end - begin - 1) steps
        return s0 + (s1 - s0) * max(it - b, 0) / max(e - b - 1, 1)

    def epoch_factor(self, epoch: int) -> float:
        if self.multistep is None:
            return 1.0
        ms, gamma = self.multistep
        return gamma ** bisect.bisect_right(ms, epoch)


# ---------------------------------------------------------------------------------------------------------
# data: synthetic batches of the reference's demo_mm_inputs shape (mmdet/testing/_utils.py:89-202)
# ---------------------------------------------------------------------------------------------------------
class SyntheticDetData:
    """``iters_per_epoch`` batches of uint8 images + 1..9 random boxes each; deterministic per (seed, epoch, index)."""

    def __init__(self, batch_size: int, num_classes: int, iters_per_epoch: int, image_hw=(800, 1333), seed: int = 0):
        self.bs, self.nc, self.n, self.hw, self.seed = batch_size, num_classes, iters_per_epoch, tuple(image_hw), seed
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self):
        return self.n

    def __iter__(self):
        H, W = self.hw
        for i in range(self.n):
            rng = np.random.RandomState((self.seed * 1000003 + self.epoch * 10007 + i) % (2 ** 31 - 1))
            imgs, samples = [], []
            for _ in range(self.bs):
                imgs.append(torch.from_numpy(rng.randint(0, 255, size=(3, H, W), dtype=np.uint8)))
                nb = rng.randint(1, 10)
                cx, cy, bw, bh = rng.rand(nb, 4).T
                boxes = np.vstack([(cx * W - W * bw / 2).clip(0, W), (cy * H - H * bh / 2).clip(0, H),
                                   (cx * W + W * bw / 2).clip(0, W), (cy * H + H * bh / 2).clip(0, H)]).T
                ds = DetDataSample(metainfo=dict(scale_factor=(1.0, 1.0), ori_shape=(H, W)))
                ds.gt_instances = InstanceData(bboxes=torch.from_numpy(boxes.astype(np.float32)),
                                               labels=torch.from_numpy(rng.randint(0, self.nc, size=nb).astype(np.int64)))
                samples.append(ds)
            yield dict(inputs=imgs, data_samples=samples)


class CocoTrainData:
    """The real-data source of `train_dataloader` (configs/_base_/datasets/coco_detection.py:37-50): CocoDataset
    annotations restricted to `metainfo.classes`, DefaultSampler(shuffle=True) sharded over ranks, AspectRatioBatchSampler,
    and the GPU image pipeline.  Batches arrive normalised and padded (`preprocessed=True`)."""

    def __init__(self, dataset_cfg: dict, batch_size: int, classes=None, scale=(1333, 800), seed: int = 0, rank: int = 0,
                 world: int = 1, device="cuda", num_workers: int = 0, prefetch_factor: int = 2):
        """num_workers / prefetch_factor: `train_dataloader.num_workers` decoding threads working `prefetch_factor`
        batches ahead of the training step (0 = decode in the training thread)"""
        from .datasets import AspectRatioBatchSampler, CocoAnnotations, GpuDetPipeline
        self.num_workers, self.prefetch_factor = int(num_workers), int(prefetch_factor)
        root = dataset_cfg.get("data_root", "")
        classes = classes or (dataset_cfg.get("metainfo") or {}).get("classes")     # None: every category of the file
        fc = dataset_cfg.get("filter_cfg") or {}
        self.ann = CocoAnnotations(os.path.join(root, dataset_cfg["ann_file"]), classes,
                                   data_prefix=os.path.join(root, (dataset_cfg.get("data_prefix") or {}).get("img", "")),
                                   filter_empty_gt=fc.get("filter_empty_gt", True), min_size=fc.get("min_size", 32))
        self.pipe = GpuDetPipeline(self.ann, scale=scale, seed=seed, device=device)
        self.bs, self.seed, self.rank, self.world, self.epoch = batch_size, seed, rank, world, 0
        self._sampler_cls = AspectRatioBatchSampler

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
        self.pipe.set_epoch(epoch)

    def _indices(self) -> List[int]:
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)                    # DefaultSampler: randperm per epoch, same on every rank
        perm = torch.randperm(len(self.ann), generator=g).tolist()
        total = -(-len(perm) // self.world) * self.world          # padded so that every rank sees the same count
        perm = (perm * (total // max(len(perm), 1) + 1))[:total]
        return perm[self.rank::self.world]

    def __len__(self):
        return -(-len(self._indices()) // self.bs)

    def __iter__(self):
        from .datasets import pinned, prefetch_map
        batches = list(self._sampler_cls(self._indices(), self.ann, self.bs))
        decode = lambda idx: (idx, [pinned(im) for im in self.pipe.decode(idx)])
        for idx, imgs in prefetch_map(decode, batches, self.num_workers, self.prefetch_factor):
            x, samples = self.pipe.assemble(idx, imgs)
            yield dict(inputs=x, data_samples=samples, preprocessed=True)


# ---------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------
def _with_next(it, prepare):
    """(item, following item or None) pairs of an iterable, every item passed through `prepare` exactly once"""
    it = iter(it)
    try:
        cur = prepare(next(it))
    except StopIteration:
        return
    for raw in it:
        nxt = prepare(raw)
        yield cur, nxt
        cur = nxt
    yield cur, None


class Runner:
    def __init__(self, cfg: Config, data=None, device: Optional[torch.device] = None, log: Callable[[str], None] = print):
        self.cfg = cfg
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.log = log if self.rank == 0 else (lambda *_: None)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.work_dir = cfg.get("work_dir") or os.path.join("work_dirs", "erd")
        model = MODELS.build(cfg.model)
        if not (cfg.get("load_from") or cfg.get("resume")):
            model.init_weights()      # backbone.init_cfg 'Pretrained' (a loaded checkpoint supersedes it); raises when unresolvable
        self.model = model.to(self.device).train()
        opt = cfg.optim_wrapper.optimizer
        if opt.type != "SGD":
            raise NotImplementedError("only SGD (the ERD configs) is built")
        if cfg.optim_wrapper.get("type", "OptimWrapper") != "OptimWrapper":
            raise NotImplementedError("AMP is not built: the path computes in fp32")
        bs = int(cfg.train_dataloader.batch_size)
        self.schedule = ParamSchedule(cfg.get("param_scheduler"))
        asl = cfg.get("auto_scale_lr") or {}
        self.trainer = ERDTrainer(self.model, lr=opt.lr, momentum=opt.get("momentum", 0.0),
                                  weight_decay=opt.get("weight_decay", 0.0),
                                  base_batch_size=asl.get("base_batch_size", 16), batch_size_per_gpu=bs,
                                  auto_scale_lr=bool(asl.get("enable", False)), warmup_iters=0)
        self.trainer.lr_factor = self.schedule.iter_factor          # warm-up comes from the config's LinearLR
        # data parallel: CUs the whole-chip grids leave to RCCL's resident kernels.  The launcher does not PROBE (ERDTrainer.
        # tune_cu_reserve spends optimisation steps on one batch: bench.py does that and prints `collectives.cu_reserve`); it takes
        # the value from ERD_CU_RESERVE, the same on every rank (tools/dist_train.sh passes the environment on)
        if self.trainer.world > 1 and os.environ.get("ERD_CU_RESERVE"):
            from . import kernels as _K
            _K.set_cu_reserve(int(os.environ["ERD_CU_RESERVE"]))
            self.trainer.cu_reserve = int(os.environ["ERD_CU_RESERVE"])
        self.max_epochs = int(cfg.train_cfg.max_epochs)
        hooks = cfg.get("default_hooks") or {}
        self.log_interval = int((hooks.get("logger") or {}).get("interval", 50))
        self.ckpt_interval = int((hooks.get("checkpoint") or {}).get("interval", 1))
        self.data = data
        self.epoch = 0
        self.history: List[Dict[str, float]] = []
        load_from, resume = cfg.get("load_from"), bool(cfg.get("resume", False))
        if resume and not load_from:
            last = os.path.join(self.work_dir, "last_checkpoint")
            if os.path.isfile(last):
                load_from = open(last).read().strip()
        if load_from:
            meta = load_checkpoint(load_from, self.model, self.trainer if resume else None)
            if resume:
                self.epoch = int(meta.get("epoch", 0))
            self.log(f"{'resumed' if resume else 'loaded'} {load_from} (epoch {self.epoch}, iter {self.trainer.iter})")

    @classmethod
    def from_cfg(cls, cfg: Config, **kw) -> "Runner":
        return cls(cfg, **kw)

    def train(self, max_iters: Optional[int] = None) -> List[Dict[str, float]]:
        pre = self.model.data_preprocessor
        if pre is None:
            raise ValueError("the config has no data_preprocessor")
        window: deque = deque(maxlen=self.log_interval)
        done = 0
        while self.epoch < self.max_epochs:
            if hasattr(self.data, "set_epoch"):
                self.data.set_epoch(self.epoch)
            self.trainer.epoch_factor = self.schedule.epoch_factor(self.epoch)
            t0 = time.perf_counter()
            for i, (out, nxt) in enumerate(_with_next(self.data, lambda b: b if b.get("preprocessed") else pre(b, True))):
                # one batch of look-ahead: the trainer queues the frozen teacher's half of the following step next to this
                # step's backward pass (ERDTrainer.train_step)
                logv = self.trainer.train_step(out["inputs"], out["data_samples"],
                                               next_batch=None if nxt is None else (nxt["inputs"], nxt["data_samples"]))
                window.append(logv)
                done += 1
                if (i + 1) % self.log_interval == 0 or (max_iters and done >= max_iters):
                    torch.cuda.synchronize()
                    keys = [k for k in window[0] if "loss" in k]
                    avg = {k: float(sum(float(w[k].detach()) for w in window) / len(window)) for k in keys}
                    # the reference's CheckInvalidLossHook (mmdet/engine/hooks/checkloss_hook.py:11-42: `isfinite(loss)`
                    # every `interval` iterations) rides on the logging read-back -- the values are on the host already
                    bad = [k for k, v in avg.items() if not math.isfinite(v)]
                    if bad:
                        raise FloatingPointError(f"loss became infinite or NaN! ({', '.join(bad)} at epoch {self.epoch + 1}, "
                                                 f"iteration {i + 1})")
                    dt = (time.perf_counter() - t0) / (i + 1)
                    rec = dict(epoch=self.epoch + 1, iter=i + 1, lr=self.trainer.last_lr, time=dt, **avg)
                    self.history.append(rec)
                    self.log("Epoch(train) [%d][%d/%d]  lr: %.4e  time: %.3f  %s" % (
                        self.epoch + 1, i + 1, len(self.data), rec["lr"], dt,
                        "  ".join(f"{k}: {v:.4f}" for k, v in avg.items())))
                if max_iters and done >= max_iters:
                    self.trainer.flush()
                    return self.history
            self.epoch += 1
            if self.rank == 0 and self.ckpt_interval > 0 and self.epoch % self.ckpt_interval == 0:
                path = os.path.join(self.work_dir, f"epoch_{self.epoch}.pth")
                save_checkpoint(path, self.model, self.trainer, meta=dict(epoch=self.epoch))
                with open(os.path.join(self.work_dir, "last_checkpoint"), "w") as f:
                    f.write(path)
                self.log(f"saved {path}")
        self.trainer.flush()
        if self.rank == 0:
            os.makedirs(self.work_dir, exist_ok=True)
            with open(os.path.join(self.work_dir, "scalars.json"), "w") as f:
                for rec in self.history:
                    f.write(json.dumps(rec) + "\n")
        return self.history
