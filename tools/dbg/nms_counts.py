#!/usr/bin/env python3
"""How many ERS-selected boxes the distillation NMS of the benched step sees per image, and what the launch costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
orig = K.distill_nms


def wrap(t_cls, t_bbox, anchors, idx_bbox, counts, iou_thr=0.005):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    s.record(st)
    out = orig(t_cls, t_bbox, anchors, idx_bbox, counts, iou_thr)
    e.record(st); torch.cuda.synchronize()
    print("ERS counts [cls, bbox] per image", counts.cpu().tolist(), "kept", out[1].cpu().tolist() if isinstance(out, tuple) else "?",
          "launch %.1f us" % (s.elapsed_time(e) * 1e3), flush=True)
    return out


K.distill_nms = wrap
for i in range(4):
    tr.train_step(*batches[i % 2])
tr.flush(); torch.cuda.synchronize()
