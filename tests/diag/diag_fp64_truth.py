#!/usr/bin/env python3
"""Full-size one-image ERD step: distance of the gradients of (a) the fp32 CPU oracle, (b) the HIP path with the teacher on
the direct kernels, (c) the HIP path with the student trunk on the Winograd kernels from an fp64 evaluation of the same step.
Answers: is the Winograd teacher less ACCURATE, or only less similar to the fp32 CPU reference's rounding pattern?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd, make_samples
from erd_amd import kernels as K
from erd_amd import parse_losses

tsd, ssd = f7_state_dicts()
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 7
imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=SEED)
x, metas = O.preprocess(imgs)
names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]


def oracle(dtype):
    t = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tsd.items()}
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in ssd.items()}
    sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    losses = O.erd_step_loss(t, sd, x.to(dtype), boxes, labels, metas, 40, 80)
    O.parse_losses(losses).backward()
    return {k: sd[k].grad.double() for k in names}, float(O.parse_losses(losses))


def gpu(wino_teacher):
    keep, K.WINO_FROZEN_TRUNK = K.WINO_FROZEN_TRUNK, wino_teacher
    try:
        model = build_erd(tsd, ssd)
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        total, _ = parse_losses(losses)
        total.backward()
        p = dict(model.named_parameters())
        return {k: p[k].grad.detach().cpu().double() for k in names}, float(total.detach())
    finally:
        K.WINO_FROZEN_TRUNK = keep


def dist(ga, gb):
    errs, num, den = [], 0.0, 0.0
    for k in names:
        a, b = ga[k], gb[k]
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        if float(b.norm()) > 1e-12:
            errs.append(float((a - b).norm() / b.norm()))
    return "median %.2e  all elements %.2e" % (float(np.median(errs)), (num / den) ** 0.5)


g64, l64 = oracle(torch.float64)
g32, l32 = oracle(torch.float32)
gd, ld = gpu(False)
gw, lw = gpu(True)
print("total loss: fp64 %.9f | fp32 CPU %.9f | HIP direct trunk %.9f | HIP Winograd trunk %.9f" % (l64, l32, ld, lw))
print("gradients vs the fp64 evaluation:")
print("  fp32 CPU oracle (the reference path)   ", dist(g32, g64))
print("  HIP, student trunk on the direct kernels     ", dist(gd, g64))
print("  HIP, student trunk on the Winograd kernels   ", dist(gw, g64))
print("gradients vs the fp32 CPU oracle:")
print("  HIP, student trunk on the direct kernels     ", dist(gd, g32))
print("  HIP, student trunk on the Winograd kernels   ", dist(gw, g32))
