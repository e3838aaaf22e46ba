#!/usr/bin/env python3
"""one training step's flat gradient with the current ERD_* env settings -> file; with two files: compare them
(bug or rounding?  a rounding-level change gives ~1e-6 relative differences with a few ReLU-flip outliers)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        d = (a[k] - b[k]).double().norm() / b[k].double().norm().clamp_min(1e-30)
        print(f"{k}: rel L2 {float(d):.3e}  max abs {float((a[k]-b[k]).abs().max()):.3e}")
    sys.exit(0)
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd, make_samples
from erd_amd.engine import ERDTrainer
tsd, ssd = f7_state_dicts()
model = build_erd(tsd, ssd)
tr = ERDTrainer(model, lr=0.0, momentum=0.0, weight_decay=0.0, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
x, metas = O.preprocess(imgs)
log = tr.train_step(x.cuda(), make_samples(boxes, labels, metas))
torch.cuda.synchronize()
out = dict(grad=tr.flat.grad.detach().cpu().clone(), loss=torch.tensor([float(log["loss"].detach())]))
with torch.no_grad():
    t_cls, t_bbox, _ = model.ori_model._forward_cat(x.cuda())
out["t_cls"], out["t_bbox"] = t_cls.cpu(), t_bbox.cpu()
torch.save(out, sys.argv[1])
