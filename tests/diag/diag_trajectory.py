#!/usr/bin/env python3
"""Three full-size optimisation steps (4 x 800x1344, alternating batches): per-step / per-entry relative deviation of the HIP
trainer's logged losses from the oracle's trajectory, for several trainer variants and learning rates, next to the ORACLE'S
OWN sensitivity (its trajectory from weights perturbed by 1e-6 relative noise)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts, build_erd, make_samples
from erd_amd.engine import ERDTrainer

tsd, ssd = f7_state_dicts()
names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
batches = []
for s in (31, 32):
    imgs, boxes, labels = O.synthetic_batch(4, 800, 1333, 40, seed=s)
    x, metas = O.preprocess(imgs)
    batches.append((x, boxes, labels, metas))
MOM, WD, STEPS = 0.9, 1e-4, 3
torch.set_num_threads(min(torch.get_num_threads(), 32))


def lrs_of(lr):
    return [lr * (0.5 + 0.5 * it / 2) for it in range(STEPS)]


def oracle_run(lr, noise=0.0):
    sd = {k: v.clone() for k, v in ssd.items()}
    if noise:
        g = torch.Generator().manual_seed(5)
        for k in names:
            sd[k] = sd[k] * (1 + noise * torch.randn(sd[k].shape, generator=g))
    bufs, ref = {}, []
    for it in range(STEPS):
        x, boxes, labels, metas = batches[it % 2]
        leaf = {k: (sd[k].clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
        losses = O.erd_step_loss(tsd, leaf, x, boxes, labels, metas, 40, 80)
        total = O.parse_losses(losses)
        total.backward()
        row = {k: float(sum(v.detach().mean() for v in vs)) for k, vs in losses.items()}
        row["loss"] = float(total)
        ref.append(row)
        O.sgd_momentum_step({k: sd[k] for k in names}, {k: leaf[k].grad for k in names}, bufs, lrs_of(lr)[it], MOM, WD)
    return ref


def hip_run(lr, ahead, share):
    os.environ["ERD_SHARE_TRUNK"] = "1" if share else "0"
    model = build_erd(tsd, ssd)
    tr = ERDTrainer(model, lr=lr, momentum=MOM, weight_decay=WD, batch_size_per_gpu=4, auto_scale_lr=False, warmup_iters=3,
                    warmup_start_factor=0.5)
    gpu = [(x.cuda(), make_samples(b, l, m)) for x, b, l, m in batches]
    logs = []
    for it in range(STEPS):
        lv = tr.train_step(*gpu[it % 2], next_batch=gpu[(it + 1) % 2] if ahead else None)
        logs.append({k: float(v) for k, v in lv.items()})
    tr.flush()
    torch.cuda.synchronize()
    assert [tr.lr_at(i) for i in range(STEPS)] == lrs_of(lr)
    return logs


def show(tag, a, b):
    for it in range(STEPS):
        print(f"  {tag} step {it}: " + "  ".join(f"{k} {abs(a[it][k] - b[it][k]) / max(abs(b[it][k]), 1e-7):.1e}" for k in b[it]))


for lr in (0.0025, 0.02):
    ref = oracle_run(lr)
    print(f"lr {lr}: oracle losses {[round(r['loss'], 6) for r in ref]}")
    show("oracle(+1e-6 noise) vs oracle", oracle_run(lr, 1e-6), ref)
    for ahead, share in ((True, True), (False, False)):
        show(f"hip ahead={ahead} share={share} vs oracle", hip_run(lr, ahead, share), ref)
