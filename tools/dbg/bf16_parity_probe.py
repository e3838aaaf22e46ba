"""HIP bf16 step vs the oracle's bf16 modes at 800x1333: gradient cosine against (a) multiplicands-only, (b) stored maps; noise floors."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O
from erd_amd import kernels as K, parse_losses
seed = int(os.environ.get("SEED", "7"))
tsd, ssd = f7_state_dicts()
names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
x, metas = O.preprocess(imgs)
torch.set_num_threads(min(torch.get_num_threads(), 32))
def oracle(ctx):
    sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in ssd.items()}
    t0 = time.time()
    if ctx is None:
        losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True); O.parse_losses(losses).backward()
    else:
        with ctx:
            losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True); O.parse_losses(losses).backward()
    print("oracle %.1fs" % (time.time() - t0), flush=True)
    return float(O.parse_losses(losses).detach()), {k: sd[k].grad.double() for k in names}, [set(aux["ers_cls"][0].tolist()), set(aux["ers_bbox"][0].tolist())]
def hip(perturb=None):
    K.set_compute("bf16")
    try:
        model = build_erd(tsd, ssd)
        if perturb:
            with torch.no_grad(): dict(model.named_parameters())[perturb].mul_(1.01)
        total, lv = parse_losses(model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")); total.backward()
        t = model.teacher_pass(x.cuda()); cnt = t.ers["counts"].cpu()
        s = [set(t.ers[n][0, :int(cnt[0, c])].cpu().tolist()) for n, c in (("idx_cls", 0), ("idx_bbox", 1))]
        p = dict(model.named_parameters())
        return float(total), {k: p[k].grad.detach().cpu().double() for k in names}, s
    finally:
        K.set_compute(K.DEFAULT_COMPUTE)
def cos(ga, gb):
    dot = na = nb = 0.0; worst = 1.0; wk = None
    for k in names:
        a, b = ga[k], gb[k]
        if float(b.norm()) < 1e-12: continue
        if b.numel() >= 4096:
            c = float((a * b).sum() / (a.norm() * b.norm()))
            if c < worst: worst, wk = c, k
        dot += float((a * b).sum()); na += float(a.pow(2).sum()); nb += float(b.pow(2).sum())
    return dot / (na * nb) ** 0.5, (na / nb) ** 0.5, worst, wk
jac = lambda A, B: ["%.3f" % (len(a & b) / max(len(a | b), 1)) for a, b in zip(A, B)]
lh, gh, sh = hip()
lp, gp, sp = hip("backbone.layer3.0.bn2.weight")
l32, g32, s32 = oracle(None)
lm, gm, sm = oracle(O.bf16_multiplicands())
ls, gs, ss = oracle(O.bf16_stored_maps())
for nm, (a, b, sa, sb, la, lb) in {"HIP vs stored": (gh, gs, sh, ss, lh, ls), "HIP vs multiplicands": (gh, gm, sh, sm, lh, lm), "stored vs fp32": (gs, g32, ss, s32, ls, l32),
                   "multiplicands vs fp32": (gm, g32, sm, s32, lm, l32), "HIP(1% bn2 scale) vs stored": (gp, gs, sp, ss, lp, ls), "HIP vs fp32": (gh, g32, sh, s32, lh, l32)}.items():
    c, r, w, wk = cos(a, b)
    print(f"{nm:30s} 1-cos {1-c:.5f} ratio {r:.4f} worst {w:.4f} ({wk}) loss rel {abs(la-lb)/abs(lb):.2e} ERS jac {jac(sa, sb)}", flush=True)
