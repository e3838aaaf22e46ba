#!/usr/bin/env python3
"""tools/parity_seeds.py --seeds 48 [--first 7] [--modes f32x3,f32] [--out profiles/r04_parity_seeds.json]

North_star's "grads within 1e-3" at BASELINE size, with enough seeds to tell chance from a shift:
for every seed one full-size ERD step (one 800x1333 image) is evaluated by
    the oracle in fp64          -- the truth
    the oracle in fp32          -- the reference's own arithmetic (torch-CPU)
    the HIP path per mode       -- "f32" (every launch on the fp32 matrix cores), "f32x3" (three-limb products, the default)
and the relative L2 distance of each fp32 evaluation's gradients from fp64 is recorded three ways: median over the 175
gradient tensors / the whole gradient (all elements) / the worst tensor.  A ReLU whose pre-activation is ~1e-7 flips under any
re-ordering of an fp32 sum and moves every upstream gradient by ~1e-3, so single seeds say nothing; the JSON carries the rows
and the summary statistics (mean, median, standard error, seeds above 1e-3, paired differences between implementations) that
tests/test_gpu_parity_full.py's assertions are sized from.  The oracle is the CHECKER here: this is a test tool (tests/diag
style), nothing under erd_amd/ imports it.  Needs a GPU (run through gpurun); the fp64 oracle takes ~8 s per seed on 32 threads."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=48)
    ap.add_argument("--first", type=int, default=7)
    ap.add_argument("--modes", default="f32x3,f32",
                    help="comma list of kernels.set_compute modes; MODE@LIB.so evaluates the mode with another build of the library "
                         "(same-process A/B of two arithmetics on the same seeds, e.g. f32x3@erd_amd/lib/abl/liberd_hip_r3trunc.so)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_seeds.json"))
    ap.add_argument("--tag", default="", help="free text stored in the JSON (e.g. which library build)")
    args = ap.parse_args()
    from oracle import erd_oracle as O
    from e2e_util import build_erd, f7_state_dicts, make_samples
    from erd_amd import kernels as K, parse_losses

    tsd, ssd = f7_state_dicts()
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
    modes = [m for m in args.modes.split(",") if m]
    torch.set_num_threads(min(torch.get_num_threads(), args.threads))

    def dist(ga, gb):
        errs, num, den = [], 0.0, 0.0
        for k in names:
            a, b = ga[k], gb[k]
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            if float(b.norm()) > 1e-12:
                errs.append(float((a - b).norm() / b.norm()))
        return [float(np.median(errs)), (num / den) ** 0.5, max(errs)]

    from erd_amd import _lib as L
    default_lib = L.LIB_PATH

    def use_lib(path):
        path = os.path.abspath(os.path.join(ROOT, path)) if not os.path.isabs(path) else path
        if path != L.LIB_PATH or L._lib is None:
            torch.cuda.synchronize()
            L.LIB_PATH, L._lib = path, None
            L.load()

    rows = {"cpu_f32": []}
    label = lambda m: "hip_" + (m.split("@")[0] + "@" + os.path.basename(m.split("@")[1]).replace("liberd_hip_", "").replace(".so", "") if "@" in m else m)
    rows.update({label(m): [] for m in modes})
    loss_dev = {label(m): 0.0 for m in modes}
    seeds = list(range(args.first, args.first + args.seeds))
    t_start = time.time()
    for seed in seeds:
        imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
        x, metas = O.preprocess(imgs)

        def oracle(dtype):
            t = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tsd.items()}
            sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in ssd.items()}
            sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
            losses = O.erd_step_loss(t, sd, x.to(dtype), boxes, labels, metas, 40, 80)
            O.parse_losses(losses).backward()
            return {k: sd[k].grad.double() for k in names}, {k: [float(v) for v in vs] for k, vs in losses.items()}

        g64, l64 = oracle(torch.float64)
        g32, _ = oracle(torch.float32)
        rows["cpu_f32"].append(dist(g32, g64))
        del g32
        for m in modes:
            use_lib(m.split("@")[1] if "@" in m else default_lib)
            K.set_compute(m.split("@")[0])
            try:
                model = build_erd(tsd, ssd)
                losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
                parse_losses(losses)[0].backward()
            finally:
                K.set_compute(K.DEFAULT_COMPUTE)
            p = dict(model.named_parameters())
            gh = {k: p[k].grad.detach().cpu().double() for k in names}
            rows[label(m)].append(dist(gh, g64))
            for k, vs in l64.items():
                got = [float(v.detach()) for v in losses[k]]
                loss_dev[label(m)] = max(loss_dev[label(m)], max(abs(a - b) / max(abs(b), 1e-7) for a, b in zip(got, vs)))
            del model, gh
        print("seed %d (%.0f s): " % (seed, time.time() - t_start) +
              " | ".join("%s %.2e %.2e %.2e" % ((k,) + tuple(v[-1])) for k, v in rows.items()), flush=True)

    def summary(a):
        a = np.array(a)
        n = len(a)
        return {"mean": [float(v) for v in a.mean(0)], "median": [float(v) for v in np.median(a, 0)],
                "sem": [float(v) for v in a.std(0, ddof=1) / np.sqrt(n)], "max": [float(v) for v in a.max(0)],
                "seeds_whole_gradient_above_1e-3": int((a[:, 1] > 1e-3).sum()),
                "seeds_whole_gradient_above_2e-3": int((a[:, 1] > 2e-3).sum())}

    out = {"what": "relative L2 distance to an fp64 evaluation of the same full-size ERD step (1 image, 800x1333); per row: "
                   "[median over the 175 gradient tensors, whole gradient, worst tensor]",
           "tag": args.tag, "seeds": seeds, "threads": torch.get_num_threads(), "rows": rows,
           "summary": {k: summary(v) for k, v in rows.items()},
           "worst_loss_entry_rel_dev_from_fp64": loss_dev, "wall_s": round(time.time() - t_start, 1)}
    # paired comparisons (same seeds): ratio of means and the paired difference with its standard error
    pairs = {}
    keys = list(rows)
    for i, a in enumerate(keys):
        for b in keys[i + 1:]:
            da = np.array(rows[a]) - np.array(rows[b])
            pairs["%s_minus_%s" % (a, b)] = {"mean": [float(v) for v in da.mean(0)],
                                             "sem": [float(v) for v in da.std(0, ddof=1) / np.sqrt(len(da))],
                                             "ratio_of_means": [float(v) for v in np.array(rows[a]).mean(0) / np.array(rows[b]).mean(0)],
                                             "seeds_a_closer": int((da[:, 1] < 0).sum())}
    out["paired"] = pairs
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["summary"].items():
        print("%-10s mean %.2e %.2e %.2e | median %.2e %.2e %.2e | sem(whole) %.1e | > 1e-3: %d of %d"
              % ((k,) + tuple(v["mean"]) + tuple(v["median"]) + (v["sem"][1], v["seeds_whole_gradient_above_1e-3"], len(seeds))))
    for k, v in pairs.items():
        print("%-28s whole-gradient mean diff %+.2e +- %.1e, ratio of means %.3f" % (k, v["mean"][1], v["sem"][1], v["ratio_of_means"][1]))


if __name__ == "__main__":
    main()
