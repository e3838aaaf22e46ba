#!/usr/bin/env python3
"""stage-by-stage comparison of the bf16 matrix-core mode against the oracle's bf16-multiplicand mode"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd
from erd_amd import kernels as K

tsd, ssd = f7_state_dicts()
imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
x, metas = O.preprocess(imgs)
for mode in ("f32", "bf16"):
    K.set_compute(mode)
    model = build_erd(tsd, ssd).eval()
    t = model.ori_model
    import contextlib
    ctx = O.bf16_multiplicands() if mode == "bf16" else contextlib.nullcontext()
    with ctx, torch.no_grad():
        feats_ref = O.resnet_forward(tsd, x)
        fpn_ref = O.fpn_forward(tsd, feats_ref)
        cls_ref, bbox_ref = O.gfl_head_forward(tsd, fpn_ref)
    with torch.no_grad():
        feats = t.backbone(x.cuda())
        fpn = t.neck(feats)
        cls, bbox = t.bbox_head(fpn)
    def rel(a, b):
        a = a.cpu()
        if a.shape != b.shape:
            a = a.permute(0, 3, 1, 2)
        return float((a - b).norm() / b.norm()), float((a - b).abs().max())
    print("mode", mode)
    for i, (a, b) in enumerate(zip(feats, feats_ref)):
        print("  C%d" % (i + 2), a.shape, rel(a, b))
    for i, (a, b) in enumerate(zip(fpn, fpn_ref)):
        print("  P%d" % (i + 3), rel(a, b))
    for i, (a, b) in enumerate(zip(cls, cls_ref)):
        print("  cls%d" % i, rel(a, b))
