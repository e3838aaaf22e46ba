"""teacher logits of the HIP bf16 mode against the oracle's bf16 modes (forward only): does `bf16_stored_maps` model the storage points?"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from e2e_util import build_erd, f7_state_dicts
from oracle import erd_oracle as O
from erd_amd import kernels as K
tsd, ssd = f7_state_dicts()
imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=7)
x, metas = O.preprocess(imgs)
torch.set_num_threads(min(torch.get_num_threads(), 32))
def orc(ctx):
    with torch.no_grad():
        if ctx is None: c, b = O.gfl_forward(tsd, x)
        else:
            with ctx: c, b = O.gfl_forward(tsd, x)
    return O.flatten_levels(c)[0], O.flatten_levels(b)[0]
K.set_compute("bf16")
try:
    model = build_erd(tsd, ssd)
    with torch.no_grad():
        t = model.teacher_pass(x.cuda())
    print("BF16_STORAGE", K.BF16_STORAGE, tuple(t.t_cls.shape), tuple(t.t_bbox.shape))
    hc, hb = t.t_cls[0].float().cpu(), t.t_bbox[0].float().cpu()
finally:
    K.set_compute(K.DEFAULT_COMPUTE)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
c32, b32 = orc(None); cm, bm = orc(O.bf16_multiplicands()); cs, bs = orc(O.bf16_stored_maps())
print("cls logits rel L2: HIP vs stored %.3e | HIP vs multiplicands %.3e | HIP vs fp32 %.3e | stored vs fp32 %.3e | mult vs fp32 %.3e | stored vs mult %.3e"
      % (rel(hc, cs), rel(hc, cm), rel(hc, c32), rel(cs, c32), rel(cm, c32), rel(cs, cm)))
print("bbox rel L2:       HIP vs stored %.3e | HIP vs multiplicands %.3e | HIP vs fp32 %.3e | stored vs fp32 %.3e | mult vs fp32 %.3e"
      % (rel(hb, bs), rel(hb, bm), rel(hb, b32), rel(bs, b32), rel(bm, b32)))
