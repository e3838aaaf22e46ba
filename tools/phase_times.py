#!/usr/bin/env python3
"""Where the un-instrumented default step spends its wall time: HIP events recorded on the step's main stream at the phase
boundaries of ERDTrainer.train_step (update, student forward, teacher join, losses, backward chain, trailing-gradient join).
usage: python tools/phase_times.py [steps] [--compute f32x3|f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
import bench
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
if "--compute" in sys.argv:
    K.set_compute(sys.argv[sys.argv.index("--compute") + 1])
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay, base_batch_size=cfg.auto_scale_lr.base_batch_size,
                batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
for j in range(4):
    tr.train_step(*batches[j % 2], next_batch=batches[(j + 1) % 2])
torch.cuda.synchronize()
ERDTrainer.PHASES = []
for j in range(4, 4 + steps):
    tr.train_step(*batches[j % 2], next_batch=batches[(j + 1) % 2])
tr.flush()
torch.cuda.synchronize()
marks = ERDTrainer.PHASES
ERDTrainer.PHASES = None
acc = collections.OrderedDict()
n = 0
for i in range(len(marks) - 1):
    (a, ea), (b, eb) = marks[i], marks[i + 1]
    key = f"{a} -> {b}"
    acc[key] = acc.get(key, 0.0) + ea.elapsed_time(eb)
    n += a == "step"
tot = sum(acc.values())
print(f"{steps} steps, {tot / n:.2f} ms per step between marks (compute {K.COMPUTE})")
for k, v in acc.items():
    print(f"  {k:48s} {v / n:7.2f} ms")
