#!/usr/bin/env python3
"""Loss curve of the ERD step on a FIXED set of synthetic batches (full size, bs=4): shows that the step optimises
(the supervised losses on the fixed batches go down) in fp32 and in the bf16 matrix-core mode, from the same start.

    python tools/convergence.py [--steps 200] [--compute f32|bf16] > profiles/rNN_convergence_<mode>.json"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--compute", default=K.DEFAULT_COMPUTE)
ap.add_argument("--nbatches", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
K.set_compute(a.compute)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay, base_batch_size=16,
                batch_size_per_gpu=4, auto_scale_lr=False, warmup_iters=50, warmup_start_factor=0.001)
batches = [bench.synthetic_gpu_batch(4, seed=100 + i, device=dev) for i in range(a.nbatches)]
curve, window = [], []
for it in range(a.steps):
    log = tr.train_step(*batches[it % a.nbatches])
    window.append({k: float(v.detach()) for k, v in log.items()})
    if (it + 1) % 20 == 0:
        avg = {k: sum(w[k] for w in window) / len(window) for k in window[0]}
        curve.append(dict(iter=it + 1, lr=tr.last_lr, **{k: round(v, 5) for k, v in avg.items()}))
        window = []
tr.flush()
ok = all(torch.isfinite(p).all().item() for p in model.parameters())
print(json.dumps(dict(compute=a.compute, steps=a.steps, batches=a.nbatches, finite=ok, curve=curve)))
