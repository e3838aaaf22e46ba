#!/usr/bin/env python3
"""which Winograd placement moves the full-size gradients away from the CPU oracle?  teacher 3x3 / student frozen trunk
(layer1) / student recorded layers / input gradients, each on the direct (D) or the Winograd (W) kernels"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd, make_samples
from erd_amd import kernels as K
from erd_amd import parse_losses

tsd, ssd = f7_state_dicts()
imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=7)
x, metas = O.preprocess(imgs)
names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in ssd.items()}
O.parse_losses(O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80)).backward()
ref = {k: sd[k].grad.double() for k in names}


def step(teacher, trunk, recorded, dgrad):
    model = build_erd(tsd, ssd)
    K.WINO_TRAIN_FWD, K.WINO_DGRAD = recorded, dgrad
    with torch.no_grad(), K.distillation_forward(teacher):
        t_cls, t_bbox, sizes = model.ori_model._forward_cat(x.cuda())
    ers = model.sel_pos_cat(t_cls, t_bbox)
    anchors = model.bbox_head.prior_generator.grid_priors_cat(sizes, t_cls.device)
    keep, _ = K.distill_nms(t_cls, t_bbox, anchors, ers["idx_bbox"], ers["counts"], 0.005)
    with K.distillation_forward(trunk):
        s_cls, s_bbox, sizes = model._forward_cat(x.cuda())
    losses = model.bbox_head.loss_cat(t_cls, t_bbox, s_cls, s_bbox, sizes, make_samples(boxes, labels, metas), ers, keep,
                                      model.ori_num_classes, model.dist_loss_weight)
    total, _ = parse_losses(losses)
    total.backward()
    p = dict(model.named_parameters())
    errs, num, den = [], 0.0, 0.0
    for k in names:
        a, b = p[k].grad.detach().cpu().double(), ref[k]
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        errs.append(float((a - b).norm() / b.norm()))
    return float(np.median(errs)), (num / den) ** 0.5


print("teacher trunk recorded dgrad | median  all-elements (vs the fp32 CPU oracle)")
for combo in itertools.product((False, True), repeat=4):
    med, glob = step(*combo)
    print("   ".join("W" if c else "D" for c in combo), "   | %.2e  %.2e" % (med, glob))
