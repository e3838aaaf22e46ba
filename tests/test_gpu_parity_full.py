"""GPU parity at BASELINE size, the parts of north_star's bar that a single seed cannot carry:

  * "ERS index masks bit-exact": a sweep over 32 full-size images, the teacher on the Winograd kernels (the default)
    and on the direct kernels, each against the CPU oracle's index sets -- mismatching images are counted, and a
    mismatch is only tolerated when it is one anchor sitting on the threshold (SURVEY R2 expects ~5e-4 such images for
    any second fp32 summation order);
  * the benched batch: N = 4 images through the whole loss (cross-image normalisers, the per-image list sums of D9) against
    the oracle, ERS sets of all four images identical;
  * "grads within 1e-3": anchored to an fp64 evaluation of the same step, over seeds 7-10 -- single ReLU decisions of
    near-zero pre-activations flip under any fp32 re-ordering and one flip moves every gradient tensor by ~1e-3 ON EITHER
    SIDE, so the distance to the fp32 CPU reference alone says nothing about who is right (the test's docstring has the
    measured table and the assertions it supports)."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("oracle_threads")]      # (32 host threads for the oracle: tests/conftest.py)

from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O


# The driver runs `pytest -m gpu` under a 1 200 s limit (<= 600 s asked): the default run takes the PINNED half of
# each sample below; ERD_TEST_FULL=1 runs the full ones (12 fp64 seeds, 32 ERS images) -- once per round by the builder
# (profiles/r05_gpu_suite_full.txt); the 144-seed statistics live in profiles/r06_parity_seeds.json either way.
FULL = os.environ.get("ERD_TEST_FULL", "0") == "1"


@pytest.fixture(scope="module")
def nets():
    tsd, ssd = f7_state_dicts()
    return tsd, ssd, build_erd(tsd, ssd)


def _nms_score_ties(t_cls, t_bbox, hw):
    """Equal-score pairs among the boxes `distill_loss_single` hands to NMS for one image (oracle values): (all pairs, pairs of one
    class whose boxes overlap by more than the NMS threshold -- the pairs whose survivor depends on the sort's tie-break)."""
    sizes = [(-(-hw[0] // s), -(-hw[1] // s)) for s in (8, 16, 32, 64, 128)]
    anchors = torch.cat(O.grid_anchors(sizes), 0)
    _, ib, _, _ = O.ers_select_single(t_cls, t_bbox)
    if ib.numel() < 2:
        return 0, 0
    conf, ids = t_cls.sigmoid().max(dim=-1)
    dec = O.distance2bbox(O.anchor_centers(anchors), O.integral(t_bbox))[ib]
    sc, cl = conf[ib], ids[ib]
    bits = sc.view(torch.int32)
    order = torch.argsort(bits)
    sb = bits[order]
    eq = (sb[1:] == sb[:-1]).nonzero().squeeze(1)
    pairs = overlapping = 0
    for e in eq.tolist():      # runs of equal scores are short (0-2 pairs per image): neighbours in sorted order cover them pairwise
        j = e
        while j >= 0 and sb[j] == sb[e + 1]:
            a, b = int(order[j]), int(order[e + 1])
            pairs += 1
            if int(cl[a]) == int(cl[b]) and float(O.bbox_overlaps(dec[a:a + 1], dec[b:b + 1], is_aligned=True)) > 0.005:
                overlapping += 1
            j -= 1
    return pairs, overlapping


def test_ers_index_sets_over_full_size_images(nets):
    from erd_amd import kernels as K
    tsd, ssd, model = nets
    model.eval()
    nimg, bs = (32 if FULL else 8), 4          # (the oracle's teacher pass on the host is 2.5 s per image)
    # teacher on Winograd (True) / direct (False) kernels, in the default fp32 form ("f32x3": direct launches on the bf16 matrix
    # cores through exact three-limb splits) and with the direct launches on the native fp32 MFMA ("f32"):
    # [images with a differing set, differing anchors]
    stats = {(True, "f32x3"): [0, 0], (False, "f32x3"): [0, 0], (True, "f32"): [0, 0], (False, "f32"): [0, 0]}
    worst_margin = 1.0
    ties = [0, 0]      # NMS tie census: [pairs of ERS-selected boxes with one score bit pattern, those of them the tie-break could touch]
    for b0 in range(0, nimg, bs):
        imgs, _, _ = O.synthetic_batch(bs, 800, 1333, 40, seed=100 + b0)
        x, _ = O.preprocess(imgs)
        with torch.no_grad():
            ref_cls, ref_bbox = O.gfl_forward(tsd, x)
        rc, rb = O.flatten_levels(ref_cls), O.flatten_levels(ref_bbox)
        for i in range(bs):
            t, o = _nms_score_ties(rc[i], rb[i], x.shape[-2:])
            ties[0] += t
            ties[1] += o
        for wino, mode in stats:
            keep, K.WINO_TEACHER = K.WINO_TEACHER, wino
            K.set_compute(mode)
            try:
                with torch.no_grad():
                    t = model.teacher_pass(x.cuda())
            finally:
                K.WINO_TEACHER = keep
                K.set_compute(K.DEFAULT_COMPUTE)
            cnt = t.ers["counts"].cpu()
            for i in range(bs):
                ic, ib, thr_c, thr_b = O.ers_select_single(rc[i], rb[i])
                gc = t.ers["idx_cls"][i, :int(cnt[i, 0])].cpu()
                gb = t.ers["idx_bbox"][i, :int(cnt[i, 1])].cpu()
                diff = set(ic.tolist()) ^ set(gc.tolist()) | set(ib.tolist()) ^ set(gb.tolist())
                if diff:
                    stats[(wino, mode)][0] += 1
                    stats[(wino, mode)][1] += len(diff)
                    # a tolerated difference is an anchor ON the threshold: its statistic within 1e-5 (relative) of it
                    mc = rc[i].sigmoid().max(-1)[0]
                    mb = rb[i].max(-1)[0]
                    for a in diff:
                        m = min(abs(float(mc[a]) - thr_c) / abs(thr_c), abs(float(mb[a]) - thr_b) / abs(thr_b))
                        worst_margin = min(worst_margin, m)
                        assert m < 1e-5, (wino, mode, b0 + i, a, m)
    print("ERS sets vs the CPU oracle over %d full-size images, images (anchors) that differ: %s"
          % (nimg, ", ".join("%s teacher / %s: %d (%d)" % ("Winograd" if w else "direct", m, v[0], v[1]) for (w, m), v in stats.items())))
    # north_star: "ERS index masks bit-exact".  The DEFAULT configuration (Winograd teacher, three-limb direct launches) must match
    # the oracle on every one of the images (measured: 0 of 32 differing in all four configurations since round 2); the three A/B
    # configurations may each own at most one image with a single anchor ON the threshold (checked above: margin < 1e-5)
    assert stats[(True, K.DEFAULT_COMPUTE)] == [0, 0], stats
    assert all(v[0] <= 1 for v in stats.values()), stats
    # mmcv's batched_nms does not promise an order among EQUAL scores (the restatement sorts stably, oracle/erd_oracle.py:600): the only
    # place that could matter is two ERS-selected boxes of one image and one class with bit-equal scores that also overlap (IoU > 0.005;
    # gfl_head_increment_erd.py:198-202).  Census over the same images: pairs with equal scores, and those that could change the keep set.
    print("NMS tie census over %d full-size images: %d equal-score pairs among the ERS-selected boxes, %d of them same-class and "
          "overlapping (the only ones an unstable sort could decide differently)" % (nimg, ties[0], ties[1]))
    assert ties[1] == 0, ties


def _cached_sensitivity():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_trajectory_sensitivity.json")) as f:
        return json.load(f)["worst_entry_rel_dev_by_step"]


def test_benched_batch_of_four_losses_and_ers_vs_oracle(nets):
    from erd_amd import parse_losses
    tsd, ssd, model = nets
    model.train()
    imgs, boxes, labels = O.synthetic_batch(4, 800, 1333, 40, seed=21)
    x, metas = O.preprocess(imgs)
    with torch.no_grad():
        ref, aux = O.erd_step_loss(tsd, ssd, x, boxes, labels, metas, 40, 80, return_aux=True)
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        t = model.teacher_pass(x.cuda())
    for k, vs in ref.items():
        got = np.array([float(v) for v in losses[k]])
        want = np.array([float(v) for v in vs])
        assert got.shape == want.shape and np.allclose(got, want, rtol=1e-3, atol=1e-7), (k, got, want)
    assert float(parse_losses(losses)[0]) == pytest.approx(float(O.parse_losses(ref)), rel=1e-4)
    cnt = t.ers["counts"].cpu()
    for i in range(4):
        assert torch.equal(t.ers["idx_cls"][i, :int(cnt[i, 0])].cpu(), aux["ers_cls"][i]), i
        assert torch.equal(t.ers["idx_bbox"][i, :int(cnt[i, 1])].cpu(), aux["ers_bbox"][i]), i


WORST_TENSOR_F32X3 = 6e-3      # pinned-seed tripwire of the three-limb form (a systematic layer error read 1.1e-2 on three seeds in a row)
SEEDS_FP64 = (7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18) if FULL else (7, 8, 9, 10, 11, 12)


def test_full_size_gradients_anchored_to_fp64(nets):
    """Six full-size steps (seeds 7-12; ERD_TEST_FULL=1: the twelve seeds 7-18 the bounds below were first written for -- every one of
    them also holds on seeds 7-12, rows of both runs in profiles/r05_gpu_suite_full.txt), each evaluated three times: the oracle in fp64 (the truth), the oracle in fp32 (the
    reference's own arithmetic) and the HIP path in both fp32 forms of its direct launches.  Relative L2 distance to fp64 as
    median over the 175 gradient tensors / all elements (= the whole gradient) / worst tensor.

    A ReLU whose pre-activation is ~1e-7 takes one side in one fp32 summation order and the other side in another; ONE such
    flip moves every gradient tensor upstream by ~1e-3 and a small-norm one (a BN bias gradient: a sum over a map with heavy
    cancellation) by up to 1e-2.  Which implementation owns a flip on a given seed is chance, and the distribution is
    heavy-tailed -- twelve seeds cannot tell chance from a 20 % shift.  The statistics that CAN are in
    profiles/r06_parity_seeds.json (tools/parity_seeds.py: 144 seeds, 7-150, same evaluation, round-6 library -- round 5's file reads the same to three digits;
    tests/test_parity_seeds_profile.py asserts what this docstring quotes from it).  Whole-gradient distance to fp64 over the 144 seeds:
                                   mean +- s.e.m.        median     seeds > 1e-3   worst seed
        cpu fp32 (the reference)   9.2e-4 +- 1.4e-4      5.7e-4     27             1.8e-2
        hip "f32" (fp32 MFMA)      6.5e-4 +- 0.4e-4      5.1e-4     26             2.8e-3
        hip "f32x3" (default)      8.8e-4 +- 1.4e-4      5.2e-4     29             1.8e-2 (seed 77: the reference's own worst seed)
        (round 4, direct launches only, seeds 7-54 also with round 3's truncating limbs: 8.6e-4 against 7.1e-4 for the round-to-nearest split on the same seeds,
         and with all NINE limb products: 8.7e-4 -- the more exact form reads further: profiles/r04_nine_products.txt)
    Both HIP forms are at least as close to fp64 as the reference's own arithmetic; between the two the paired difference over the
    144 seeds is +2.3e-4 +- 1.4e-4 (1.6 standard errors; the medians are equal): not distinguishable.  On random 12-seed subsets of
    the first 48 seeds the ratio of the two forms' means ranges from 0.6 to 2.4 (90th percentile 1.48), and the three-limb form
    has more than one seed above 1e-3 beyond the reference's count on 11 % of the subsets -- so the assertions here, on the
    PINNED seeds 7-18, are (values measured on these seeds with the direct launches' limb split alone in brackets; the Winograd
    launches have since joined the three-limb form, which redraws the lottery of which ReLU flips on which seed):
      A. every seed, both forms: losses within 1e-3 of the fp32 reference and of fp64.
         "f32": the whole gradient within 1e-3 of fp64 on EVERY seed -- the original bound [worst 8.9e-4].
         "f32x3": its own, named bound: the number of seeds above 1e-3 at most the reference's own count + 1 [1 vs 1], no seed
         above 2.5e-3 [worst 1.2e-3], the median seed within 1e-3;
      B. means over the seeds (whole gradient, per-tensor median, worst tensor) at most 1.25 x the reference's own [f32x3:
         1.14 / 1.00 / 1.20; f32: 0.95 / 0.93 / 1.03], and the three-limb mean at most 1.5 x the native form's [1.21; 1.5 is the
         90th percentile of that ratio over 12-seed subsets of the 48 -- the bound a 12-seed sample supports; "<= 1.1" is asserted
         where it can be, on the 48-seed medians in test_parity_seeds_profile.py];
      C. worst single tensor <= 6e-3 on every pinned seed [f32x3 worst 4.98e-3 on seed 13, f32 3.5e-3; the reference's own
         4.8e-3] -- a systematic error in one layer (the Winograd double store of round 2 read 1.1e-2 on three seeds in a row)
         shows up in this column first.  Over the 48 seeds single flips put it at up to 8.3e-3 for both HIP forms and 1.3e-2
         for the reference, so this is a pinned-seed tripwire, not a statistical statement; B's worst-tensor MEAN is the latter.
    Round 5 (every product three-limb, Winograd included; seeds 7-18, profiles/r05_gpu_suite_full.txt): A f32x3 1 seed above 1e-3 vs
    the reference's 1, worst 1.14e-3; B means / the reference's 0.93 / 1.01 / 1.03 (f32: 0.93 / 0.95 / 1.03), three-limb / native
    1.07; C worst tensor 4.2e-3 (f32: 3.5e-3)."""
    from erd_amd import parse_losses
    tsd, ssd, _ = nets
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))        # (oneDNN oversubscribes on a 128-core host: bench.py's thread sweep)

    def dist(ga, gb):
        errs, num, den = [], 0.0, 0.0
        for k in names:
            a, b = ga[k], gb[k]
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            if float(b.norm()) > 1e-12:
                errs.append(float((a - b).norm() / b.norm()))
        return float(np.median(errs)), (num / den) ** 0.5, max(errs)

    from erd_amd import kernels as K
    MODES = ("f32x3", "f32")       # the default fp32 form and the native fp32-MFMA form of the direct launches
    cpu_rows, hip_rows = [], {m: [] for m in MODES}
    try:
        def sample(seed):
            imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
            x, metas = O.preprocess(imgs)
            return x, boxes, labels, metas

        def oracle(seed, dtype):
            x, boxes, labels, metas = sample(seed)
            t = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in tsd.items()}
            sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in ssd.items()}
            sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
            losses = O.erd_step_loss(t, sd, x.to(dtype), boxes, labels, metas, 40, 80)
            O.parse_losses(losses).backward()
            return {k: sd[k].grad.double() for k in names}, {k: [float(v) for v in vs] for k, vs in losses.items()}

        for seed in SEEDS_FP64:
            x, boxes, labels, metas = sample(seed)
            g64, l64 = oracle(seed, torch.float64)
            g32, l32 = oracle(seed, torch.float32)
            cpu = dist(g32, g64)
            cpu_rows.append(cpu)
            for mode in MODES:
                K.set_compute(mode)
                try:
                    model = build_erd(tsd, ssd)
                    losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
                    parse_losses(losses)[0].backward()
                finally:
                    K.set_compute(K.DEFAULT_COMPUTE)
                p = dict(model.named_parameters())
                gh = {k: p[k].grad.detach().cpu().double() for k in names}
                for k, vs in l32.items():
                    got = [float(v.detach()) for v in losses[k]]
                    assert np.allclose(got, vs, rtol=1e-3, atol=1e-7) and np.allclose(got, l64[k], rtol=1e-3, atol=1e-7), (seed, mode, k, got, vs)
                hip = dist(gh, g64)
                hip_rows[mode].append(hip)
                print("seed %d, %d gradient tensors, rel L2 to fp64 (median / all elements / worst tensor): cpu fp32 %.2e %.2e %.2e | hip %s %.2e %.2e %.2e"
                      % ((seed, len(names)) + cpu + (mode,) + hip))
                del model
    finally:
        torch.set_num_threads(threads)
    cpu = np.array(cpu_rows)
    for mode in MODES:
        hip = np.array(hip_rows[mode])
        print("means over %d seeds (median / whole gradient / worst tensor): cpu fp32 %.2e %.2e %.2e | hip %s %.2e %.2e %.2e"
              % ((len(SEEDS_FP64),) + tuple(cpu.mean(0)) + (mode,) + tuple(hip.mean(0))))
        print("seeds with the whole gradient more than 1e-3 from fp64: cpu fp32 %d, hip %s %d (of %d)"
              % (int((cpu[:, 1] > 1e-3).sum()), mode, int((hip[:, 1] > 1e-3).sum()), len(SEEDS_FP64)))
        above, above_cpu = int((hip[:, 1] > 1e-3).sum()), int((cpu[:, 1] > 1e-3).sum())
        if mode == "f32":                                                                            # A, the original bound
            assert (hip[:, 1] <= 1e-3).all(), (mode, hip[:, 1])
        else:                                                                                        # A, the three-limb form's own
            assert above <= above_cpu + 1 and (hip[:, 1] <= 2.5e-3).all() and float(np.median(hip[:, 1])) <= 1e-3, (mode, hip[:, 1], cpu[:, 1])
        assert (hip.mean(0) <= 1.25 * cpu.mean(0)).all(), (mode, hip.mean(0), cpu.mean(0))          # B
        # C: 5e-3 for the native form as in round 3 [3.5e-3]; the three-limb form keeps its own, named bound [4.98e-3 on seed 13]
        assert (hip[:, 2] <= (5e-3 if mode == "f32" else WORST_TENSOR_F32X3)).all(), (mode, hip[:, 2])
    x3, f32 = np.array(hip_rows["f32x3"]), np.array(hip_rows["f32"])
    print("three-limb / native means over the pinned seeds (median, whole gradient, worst tensor): %s" % (x3.mean(0) / f32.mean(0),))
    assert x3[:, 1].mean() <= 1.5 * f32[:, 1].mean(), (x3[:, 1].mean(), f32[:, 1].mean())             # B, between the two forms


def test_benched_trainer_configuration_follows_the_oracle_trajectory_at_full_size(nets, monkeypatch):
    """The configuration bench.py times -- ERDTrainer with every default on (teacher look-ahead on the side stream, trailing
    weight-gradient stream, per-step parameter preparation, batched BN fold, shared frozen trunk, cls || reg towers) at
    BASELINE's batch (4 images of 800x1344) -- against the oracle as a WHOLE: three optimisation steps on two alternating
    batches (gfl_increment_erd.py:202-220 + the optimizer wrapper's SGD).
      * per-step losses against the oracle's trajectory (the oracle applies its own SGD between steps): the total within 1e-3 at
        every step, every entry within 1e-3 while the problem is conditioned that well (see the note at the assertion);
      * parameter displacement after the third update against the oracle's (relative L2 over all trainable tensors);
      * the first step is bit-identical from run to run (deterministic forward reductions);
      * the same trajectory with the shared trunk off / without the look-ahead / with both off, to float-atomic-order noise:
        the overlap machinery changes WHEN kernels run, never what they compute.
    (The trainer-level tests in test_gpu_e2e.py run at 123x153, where the five streams barely overlap; the Winograd double
    store of round 2 only showed at full size.)"""
    from erd_amd.engine import ERDTrainer
    tsd, ssd, _ = nets
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
    batches = []
    for s in (31, 32):
        imgs, boxes, labels = O.synthetic_batch(4, 800, 1333, 40, seed=s)
        x, metas = O.preprocess(imgs)
        batches.append((x, boxes, labels, metas))
    # the learning rate of the shipped schedule for this batch: optimizer lr 0.01 x auto_scale_lr 4 / 16 (config :112-116),
    # with a 3-iteration warm-up from 0.5 so that the schedule code is on the path.  (At 8x this rate two updates amplify the
    # 5e-4 gradient noise of ANY second fp32 implementation -- test_full_size_gradients_anchored_to_fp64 -- past 1e-3 in
    # `loss_dist_bbox`, a difference of two nearly equal responses: observed 1.9e-3 at lr 0.02.)
    LR, MOM, WD, STEPS = 0.0025, 0.9, 1e-4, 3

    def hip_run(ahead: bool, share: bool):
        if share:
            monkeypatch.delenv("ERD_SHARE_TRUNK", raising=False)
        else:
            monkeypatch.setenv("ERD_SHARE_TRUNK", "0")
        model = build_erd(tsd, ssd)
        assert model.shares_trunk() == share
        tr = ERDTrainer(model, lr=LR, momentum=MOM, weight_decay=WD, batch_size_per_gpu=4, auto_scale_lr=False,
                        warmup_iters=3, warmup_start_factor=0.5)
        assert tr.prep is not None and tr.prefold is not None and tr.overlap_teacher      # the defaults ARE on
        gpu = [(x.cuda(), make_samples(b, l, m)) for x, b, l, m in batches]
        logs = []
        for it in range(STEPS):
            nxt = gpu[(it + 1) % 2] if ahead else None
            lv = tr.train_step(*gpu[it % 2], next_batch=nxt)
            if ahead:
                assert tr._teacher_ahead is not None
            logs.append({k: float(v) for k, v in lv.items()})
        tr.flush()
        torch.cuda.synchronize()
        if ahead:
            assert len(tr.prep.recipes) > 50                                                # prepared buffers were in use
        p = dict(model.named_parameters())
        return logs, {k: p[k].detach().cpu() for k in names}, [tr.lr_at(i) for i in range(STEPS)]

    logs, params, lrs = hip_run(True, True)
    # ---- the oracle's trajectory, and the oracle's trajectory from weights perturbed by 1e-6 (relative): its own conditioning
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))

    def oracle_run(noise: float):
        sd = {k: v.clone() for k, v in ssd.items()}
        if noise:
            gen = torch.Generator().manual_seed(5)
            for k in names:
                sd[k] = sd[k] * (1 + noise * torch.randn(sd[k].shape, generator=gen))
        bufs, rows = {}, []
        for it in range(STEPS):
            x, boxes, labels, metas = batches[it % 2]
            leaf = {k: (sd[k].clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
            losses = O.erd_step_loss(tsd, leaf, x, boxes, labels, metas, 40, 80)
            total = O.parse_losses(losses)
            total.backward()
            row = {k: float(sum(v.detach().mean() for v in vs)) for k, vs in losses.items()}
            row["loss"] = float(total.detach())
            rows.append(row)
            O.sgd_momentum_step({k: sd[k] for k in names}, {k: leaf[k].grad for k in names}, bufs, lrs[it], MOM, WD)
            del leaf, losses, total
        return rows, sd

    try:
        ref, sd = oracle_run(0.0)
        # (the oracle's own 1e-6 sensitivity: a second trajectory on the host, 25 s -- ERD_TEST_FULL=1; the default run uses the value
        #  it has read since round 3, 1.7e-3 at the third step)
        ref_eps = oracle_run(1e-6)[0] if FULL else None
    finally:
        torch.set_num_threads(threads)
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-7)
    for it in range(STEPS):
        print("step %d, relative deviation from the oracle's trajectory, hip | the oracle itself from weights x (1 + 1e-6 noise): %s"
              % (it, "  ".join("%s %.1e | %s" % (k, rel(logs[it][k], v), "%.1e" % rel(ref_eps[it][k], v) if ref_eps else "-") for k, v in ref[it].items())))
    # Measured (lr 0.0025): every entry within 2.4e-5 / 2.2e-4 of the oracle at steps 0 / 1, the TOTAL within 7e-5 at step 2;
    # single entries at step 2 up to 1.4e-3 (loss_dist_cls) -- where the oracle's OWN trajectory from weights perturbed by
    # 1e-6 is 1.7e-3 away: after two updates the per-image distillation terms (a few hundred ERS anchors each) are conditioned
    # no better than that for ANY implementation.  Asserted: total loss within 1e-3 at every step; every entry within 1e-3 at
    # steps 0 and 1; at step 2 within the larger of 1e-3 and twice the oracle's own 1e-6 sensitivity of its WORST entry at that
    # step (round 4: with the rounding limb split the deviation moved to loss_dist_bbox, 1.3e-3, where this noise realisation of
    # the oracle happened to move little -- which entry a perturbation lands on is chance, the step's conditioning is not).
    for it, (g, r) in enumerate(zip(logs, ref)):
        assert g["loss"] == pytest.approx(r["loss"], rel=1e-3), (it, g["loss"], r["loss"])
        # the oracle's own sensitivity at this step: MEASURED in the full run (and held against the committed value: the cache cannot rot
        # silently), read from tests/golden/oracle_trajectory_sensitivity.json in the default run (the full run of round 6 wrote it)
        cached = _cached_sensitivity()[it]
        if ref_eps:
            own = max(rel(ref_eps[it][k], v) for k, v in r.items())
            print("step %d: the oracle's own worst-entry sensitivity to 1e-6 weight noise %.2e (committed %.2e)" % (it, own, cached))
            if it == 2:
                assert 0.4 * cached <= own <= 2.5 * cached, (own, cached)
        else:
            own = cached
        for k, v in r.items():
            tol = 1e-3 if it < 2 else max(1e-3, 2.0 * own)
            assert rel(g[k], v) <= tol, (it, k, g[k], v, tol)
    num = sum(float((params[k].double() - ssd[k].double() - (sd[k].double() - ssd[k].double())).pow(2).sum()) for k in names)
    den = sum(float((sd[k].double() - ssd[k].double()).pow(2).sum()) for k in names)
    disp = (num / den) ** 0.5
    print("benched configuration, 3 steps at 4 x 800x1344: losses %s vs oracle %s; displacement rel L2 error %.2e"
          % ([round(l["loss"], 6) for l in logs], [round(r["loss"], 6) for r in ref], disp))
    assert disp < 1e-2, disp
    # ---- run to run, and the overlap machinery switched off piece by piece
    again, _, _ = hip_run(True, True)
    assert again[0] == logs[0]                                                             # bit-identical first step
    for ahead, share in ((False, True), (True, False), (False, False)):
        other, oparams, _ = hip_run(ahead, share)
        for it, (a, b) in enumerate(zip(logs, other)):
            for k in a:      # (without the shared trunk the teacher's trunk runs on the Winograd kernels: 1e-6-level logit changes;
                # from the second step on such a change now and then moves ONE anchor across an ERS threshold or an ATSS tie -- round 5
                # read 5.5e-4 in loss_cls at step 1 on one box in two runs -- so later steps get the 1e-3 of the oracle comparison above)
                assert b[k] == pytest.approx(a[k], rel=(5e-6 if share else 1e-4) if it == 0 else 1e-3, abs=1e-6), (ahead, share, it, k, a[k], b[k])
        n2 = sum(float((oparams[k].double() - params[k].double()).pow(2).sum()) for k in names)
        d2 = sum(float((params[k].double() - ssd[k].double()).pow(2).sum()) for k in names)
        assert (n2 / d2) ** 0.5 < 5e-3, (ahead, share, (n2 / d2) ** 0.5)
