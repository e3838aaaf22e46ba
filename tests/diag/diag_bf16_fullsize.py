"""bf16 mode (bf16-stored maps) against the fp32 HIP path at BASELINE size: losses, ERS sets, gradients.  Prints what the
bounds of test_gpu_e2e.py::test_bf16_full_size_step_against_the_fp32_path are taken from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O
from erd_amd import kernels as K, parse_losses

tsd, ssd = f7_state_dicts()
for seed in (7, 8):
    imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
    x, metas = O.preprocess(imgs)
    out = {}
    for mode, storage in (("f32", True), ("bf16", True), ("bf16", False)):
        K.set_compute(mode); K.BF16_STORAGE = storage
        model = build_erd(tsd, ssd)
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        total, lv = parse_losses(losses)
        total.backward()
        tc, tb, sizes = model.ori_model._forward_cat(x.cuda())
        ers = model.sel_pos_cat(tc, tb)
        cnt = ers["counts"].cpu()
        sets = [set(ers[n][0, :int(cnt[0, c])].cpu().tolist()) for n, c in (("idx_cls", 0), ("idx_bbox", 1))]
        g = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
        out[(mode, storage)] = ({k: float(v) for k, v in lv.items()}, g, sets)
        del model
    K.set_compute("f32"); K.BF16_STORAGE = True
    ref = out[("f32", True)]
    for key in (("bf16", True), ("bf16", False)):
        l, g, sets = out[key]
        lerr = {k: abs(l[k] - ref[0][k]) / max(abs(ref[0][k]), 1e-9) for k in l}
        errs, cos, num, den, dot, na, nb = [], [], 0.0, 0.0, 0.0, 0.0, 0.0
        for k, b in ref[1].items():
            a = g[k]
            if float(b.norm()) < 1e-12: continue
            errs.append(float((a - b).norm() / b.norm()))
            cos.append(float((a * b).sum() / (a.norm() * b.norm() + 1e-300)))
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            dot += float((a * b).sum()); na += float(a.pow(2).sum()); nb += float(b.pow(2).sum())
        jac = [len(a & b) / max(len(a | b), 1) for a, b in zip(sets, ref[2])]
        print(f"seed {seed} {key}: losses max rel {max(lerr.values()):.3e} ({max(lerr, key=lerr.get)}), total rel {lerr.get('loss', 0):.3e}; "
              f"grad rel L2 median {np.median(errs):.3e} max {max(errs):.3e} global {(num/den)**0.5:.3e}; cos min {min(cos):.4f} median {np.median(cos):.4f} "
              f"global {dot/(na*nb)**0.5:.5f}; norm ratio {(na/nb)**0.5:.4f}; ERS jaccard {jac} sizes {[len(s) for s in ref[2]]}")
