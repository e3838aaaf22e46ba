#!/usr/bin/env python3
"""Per-layer-shape time table of the GEMM-shaped launches of one serialized ERD step (HIP events per launch):
which shapes own the step, at what TFLOP/s, against max(flop / 157.3 TF, bytes / 5 TB/s).
usage: python tools/step_breakdown.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from erd_amd import functional as Fn
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
tr.overlap_teacher = False
Fn.TOWERS_ON_TWO_STREAMS = False
Fn.WGRAD_TRAIL = False          # (trailing weight gradients would overlap the input-gradient launches being timed)
for i in range(2):
    tr.train_step(*batches[i % 2])
tr.flush(); torch.cuda.synchronize()
K.TIMING_DETAIL = True
K.timing_begin()
for i in range(steps):
    tr.train_step(*batches[i % 2])
tr.flush()
rec = K.timing_end()
rows = sorted(rec.values(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows) / steps
print(f"{'kernel / shape':62s} {'n/step':>6s} {'ms/step':>8s} {'us/launch':>9s} {'TF':>6s} {'bound us':>8s} {'x bound':>7s}")
for r in rows:
    n = r["launches"] / steps
    us = 1e3 * r["ms"] / r["launches"]
    fl, by = r["flop"] / r["launches"], r["min_bytes"] / r["launches"]
    bound = max(fl / 157.3e12, by / 5e12) * 1e6
    print(f"{r['kernel']:62s} {n:6.1f} {r['ms'] / steps:8.3f} {us:9.1f} {fl / us / 1e6:6.1f} {bound:8.1f} {us / bound:7.2f}")
print(f"total {tot:.2f} ms/step over {len(rows)} shapes")
