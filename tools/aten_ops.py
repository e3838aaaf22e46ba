#!/usr/bin/env python3
"""Which torch (ATen) operators still launch device work inside one default training step, and from where: one step under
torch.profiler (CPU activity, shapes + Python stacks), grouped by operator, input shape and the innermost erd_amd / bench frame.
The arithmetic of the path lives in liberd_hip.so; what shows up here is glue (autograd fan-in sums, zero fills, clones).
usage: python tools/aten_ops.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from erd_amd.engine import ERDTrainer

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay, base_batch_size=cfg.auto_scale_lr.base_batch_size,
                batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
for j in range(4):
    tr.train_step(*batches[j % 2], next_batch=batches[(j + 1) % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.train_step(*batches[0], next_batch=batches[1])
    tr.flush()
    torch.cuda.synchronize()
WATCH = ("aten::add", "aten::add_", "aten::copy_", "aten::zero_", "aten::fill_", "aten::sum", "aten::mul", "aten::clone", "aten::cat",
         "aten::stack", "aten::index", "aten::select_backward", "aten::slice_backward", "aten::unbind", "aten::mean", "aten::sub", "aten::div",
         "aten::cumsum", "aten::to", "aten::_to_copy", "aten::index_put_", "aten::masked_fill_", "aten::all", "aten::eq", "aten::ne")
agg = collections.Counter()
for e in prof.events():
    if e.name not in WATCH:
        continue
    shapes = str([tuple(s) for s in (e.input_shapes or []) if s])[:70]
    frame = next((f for f in (e.stack or []) if ("erd_amd" in f or "bench.py" in f) and "torch/" not in f), (e.stack or ["?"])[0] if e.stack else "autograd engine (no Python frame)")
    agg[(e.name, shapes, frame.split("/repo/")[-1][:90])] += 1
for (name, shapes, frame), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:4d}  {name:22s} {shapes:72s} {frame}")
