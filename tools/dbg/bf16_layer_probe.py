"""where does the HIP bf16 mode round differently from the oracle's `bf16_stored_maps`?  stage outputs, one by one"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from e2e_util import build_erd, f7_state_dicts
from oracle import erd_oracle as O
from erd_amd import kernels as K
tsd, ssd = f7_state_dicts()
imgs, boxes, labels = O.synthetic_batch(1, 400, 600, 40, seed=7)
x, metas = O.preprocess(imgs)
torch.set_num_threads(min(torch.get_num_threads(), 32))
def stats(name, h, o):
    h, o = h.float().cpu(), o.float()
    d = (h - o)
    nz = (d != 0).float().mean()
    print(f"{name:14s} rel L2 {float(d.double().norm() / o.double().norm()):.3e}   elements that differ {float(nz):.4f}   max |d|/max|o| {float(d.abs().max() / o.abs().max()):.3e}", flush=True)
K.set_compute("bf16")
try:
    model = build_erd(tsd, ssd)
    with torch.no_grad():
        outs = model.backbone(x.cuda())
        fp = model.neck(outs)
        cls, bbox = model.bbox_head(fp)
finally:
    K.set_compute(K.DEFAULT_COMPUTE)
for nm, ctx in (("stored", O.bf16_stored_maps()), ("multiplicands", O.bf16_multiplicands())):
    print("== oracle mode:", nm)
    with torch.no_grad(), ctx:
        ro = O.resnet_forward(ssd, x)
        rf = O.fpn_forward(ssd, ro)
        rc, rb = O.gfl_head_forward(ssd, rf)
    for i in range(4): stats(f"C{i+2}", outs[i], ro[i])
    for i in range(5): stats(f"P{i+3}", fp[i], rf[i])
    for i in range(5): stats(f"cls{i}", cls[i], rc[i])
    for i in range(5): stats(f"reg{i}", bbox[i], rb[i])
