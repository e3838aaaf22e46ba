#!/usr/bin/env python3
"""host (launch) time per training step vs wall time: is the step GPU-bound or launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from erd_amd.engine import ERDTrainer

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
for i in range(3):
    tr.train_step(*batches[i % 2])
tr.flush(); torch.cuda.synchronize()
n = 8
host = 0.0
t0 = time.perf_counter()
for i in range(n):
    h0 = time.perf_counter()
    tr.train_step(*batches[i % 2])
    host += time.perf_counter() - h0
tr.flush()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"per step: host (issue) {1e3 * host / n:.1f} ms, wall {1e3 * wall / n:.1f} ms; issue loop finished after {1e3 * t_issue / n:.1f} ms/step")
