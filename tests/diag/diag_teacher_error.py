#!/usr/bin/env python3
"""teacher head outputs at full size against an fp64 forward: direct vs Winograd 3x3 kernels, per level"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd
from erd_amd import kernels as K

tsd, ssd = f7_state_dicts()
imgs, _, _ = O.synthetic_batch(1, 800, 1333, 40, seed=7)
x, _ = O.preprocess(imgs)
with torch.no_grad():
    c64, b64 = O.gfl_forward({k: (v.double() if v.is_floating_point() else v) for k, v in tsd.items()}, x.double())
    c32, b32 = O.gfl_forward(tsd, x)
model = build_erd(tsd, ssd).eval()
sizes = [tuple(m.shape[-2:]) for m in c64]
def per_level(t_cat, refs):
    out, off = [], 0
    for r in refs:
        n = r.shape[2] * r.shape[3]
        a = t_cat[0, off:off + n].cpu().double()
        b = r[0].permute(1, 2, 0).reshape(n, -1)
        out.append(float((a - b).norm() / b.norm())); off += n
    return " ".join("%.2e" % v for v in out)
print("torch-CPU fp32 vs fp64   cls:", " ".join("%.2e" % float((a.double() - b).norm() / b.norm()) for a, b in zip(c32, c64)),
      "| bbox:", " ".join("%.2e" % float((a.double() - b).norm() / b.norm()) for a, b in zip(b32, b64)))
for wino in (False, True):
    K.WINO_TEACHER = wino
    with torch.no_grad():
        t = model.teacher_pass(x.cuda())
    print("HIP teacher, %-8s 3x3  cls:" % ("Winograd" if wino else "direct"), per_level(t.t_cls, c64), "| bbox:", per_level(t.t_bbox, b64))
