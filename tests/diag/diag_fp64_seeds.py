#!/usr/bin/env python3
"""per seed: median / global / per-tensor-max relative L2 distance of the gradients of (a) the fp32 CPU oracle and (b) the
HIP path from an fp64 evaluation of the same full-size step, and of (b) from (a); wall times of the three evaluations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd, make_samples
from erd_amd import parse_losses

tsd, ssd = f7_state_dicts()
names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]


def oracle(dtype, x, boxes, labels, metas):
    t = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tsd.items()}
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in ssd.items()}
    sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    losses = O.erd_step_loss(t, sd, x.to(dtype), boxes, labels, metas, 40, 80)
    O.parse_losses(losses).backward()
    return {k: sd[k].grad.double() for k in names}


def dist(ga, gb):
    errs, num, den = {}, 0.0, 0.0
    for k in names:
        a, b = ga[k], gb[k]
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        if float(b.norm()) > 1e-12:
            errs[k] = float((a - b).norm() / b.norm())
    worst = max(errs, key=errs.get)
    return float(np.median(list(errs.values()))), (num / den) ** 0.5, errs[worst], worst


for seed in [int(a) for a in sys.argv[1:]] or [7, 8, 9, 10]:
    imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
    x, metas = O.preprocess(imgs)
    t0 = time.time(); g64 = oracle(torch.float64, x, boxes, labels, metas); t64 = time.time() - t0
    t0 = time.time(); g32 = oracle(torch.float32, x, boxes, labels, metas); t32 = time.time() - t0
    model = build_erd(tsd, ssd)
    total, _ = parse_losses(model(x.cuda(), make_samples(boxes, labels, metas), mode="loss"))
    total.backward()
    p = dict(model.named_parameters())
    gg = {k: p[k].grad.detach().cpu().double() for k in names}
    print(f"seed {seed}: fp64 {t64:.0f} s, fp32 {t32:.0f} s")
    for tag, a, b in (("cpu32 vs fp64", g32, g64), ("hip   vs fp64", gg, g64), ("hip   vs cpu32", gg, g32)):
        med, glob, mx, worst = dist(a, b)
        print(f"   {tag}: median {med:.2e} global {glob:.2e} max {mx:.2e} ({worst})")
    del model
