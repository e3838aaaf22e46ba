"""GPU end-to-end parity: the full ERD step (teacher fwd -> ERS -> NMS -> student fwd -> 5 loss groups ->
backward) through the registry/config boundary, against
  * F7: losses + per-parameter gradient norms produced by the REAL reference (ResNet-50, 128x160), and
  * the oracle restatement run live on CPU at a second size (ragged image sizes, padded anchors).

Tolerances.  Losses / activations: 1e-3 relative (north_star; observed ~1e-5).  Gradients: every fused
op's backward is held to <=2e-4 against torch autograd in test_gpu_functions.py.  End to end, two fp32
implementations of a ReLU network cannot agree elementwise to 1e-3 on a 128x160 image: a pre-activation
within ~1e-6 of zero flips its ReLU mask (measured: 2 of 8M activations flip GPU-vs-CPU) and one flipped
pixel moves every upstream weight gradient by ~1 % at this size (the reference run twice on CPU with a
2e-6 input perturbation shows the same 5e-3 jumps: tests/test_oracle_sensitivity.py).  So end to end we
assert: median per-tensor relative L2 error < 1e-3 (F7: norm error < 1e-4), every tensor < 5e-2, whole-gradient L2 error < 2e-2."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("oracle_threads")]      # (32 host threads for the oracle: tests/conftest.py)

import golden_inputs as G
from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O

RTOL = 1e-3


def _lossdict_to_np(d):
    return {k: np.array([float(v) for v in vs], dtype=np.float64) for k, vs in d.items()}


@pytest.mark.parametrize("fixture,c_old,depth", [("f7_tiny_e2e.npz", 40, 50), ("f9_tiny_e2e_r101_70_10.npz", 70, 101)])
def test_f7_reference_fixture_losses_and_grads(golden, fixture, c_old, depth):
    """F7: GFL-R50 40+40 (BASELINE configs[1]); F9: GFL-R101 70+10 (configs[3]: deeper backbone, a 70-channel teacher
    head -- rows that are not 16-B aligned -- and only 10 new classes)."""
    from erd_amd import parse_losses
    import e2e_util as U
    g = golden(fixture)
    tsd, ssd = f7_state_dicts(c_old, 80, depth)
    model = build_erd(tsd, ssd) if depth == 50 else build_erd(tsd, ssd, cfg_first=U.CFG_FIRST70, cfg_incre=U.CFG_INCRE10)
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 80 - c_old, seed=0)
    x, metas = O.preprocess(imgs)
    # teacher outputs (mode='tensor' API, NCHW views)
    t_cls, t_bbox = model.ori_model(x.cuda(), mode="tensor")
    assert np.allclose(t_cls[0][0, :, ::4, ::4].cpu().numpy(), g["teacher_cls0_sample"], rtol=RTOL, atol=1e-4)
    assert np.allclose(t_bbox[4].cpu().numpy(), g["teacher_bbox4"], rtol=RTOL, atol=1e-4)
    losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
    L = _lossdict_to_np(losses)
    for k in ("loss_cls", "loss_bbox", "loss_dfl", "loss_dist_cls", "loss_dist_bbox"):
        assert np.allclose(L[k], g[k], rtol=RTOL, atol=1e-7), (k, L[k], g[k])
    total, _ = parse_losses(losses)
    assert float(total) == pytest.approx(float(g["total"]), rel=RTOL)
    total.backward()
    names = [str(n) for n in g["grad_names"]]
    params = dict(model.named_parameters())
    got = sorted(k for k, p in params.items() if p.requires_grad and not k.startswith("ori_model."))
    assert got == sorted(names)
    errs = []
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        ref = float(g["grad_norms"][i])
        if ref < 1e-12:
            assert float(gr.double().norm()) < 1e-9, k
            continue
        errs.append(abs(float(gr.double().norm()) - ref) / ref)
        assert errs[-1] < 5e-2, (k, float(gr.double().norm()), ref)
    # the R101 70+10 step is ~16x worse conditioned than R50 40+40 (twice the ReLUs, 10 new classes, few positives):
    # the oracle itself moves by 2.7e-4 (median grad-norm) under a 2e-6 input perturbation, 1.7e-5 for R50
    # (tests/test_oracle_sensitivity.py)
    assert float(np.median(errs)) < (1e-4 if depth == 50 else 2e-3), float(np.median(errs))
    print("grad-norm rel err: median %.2e max %.2e" % (float(np.median(errs)), max(errs)))
    # teacher is frozen
    assert all(p.grad is None for k, p in params.items() if k.startswith("ori_model."))


def test_live_oracle_ragged_batch():
    """different image sizes in one batch (pad to /32), N=3: losses, ERS index sets and all gradients vs oracle."""
    from erd_amd import parse_losses
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd)
    rng = np.random.RandomState(5)
    imgs = [torch.from_numpy(rng.randint(0, 255, size=(3, h, w), dtype=np.uint8)) for h, w in [(150, 200), (131, 217), (160, 180)]]
    boxes = [G.rand_boxes(900 + i, 2 + i, 170.0, 125.0, min_size=10.0) for i in range(3)]
    labels = [G.randint(910 + i, 0, 40, 2 + i) for i in range(3)]
    x, metas = O.preprocess(imgs)
    sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v) for k, v in ssd.items()}
    ref_losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
    ref_total = O.parse_losses(ref_losses)
    ref_total.backward()
    losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
    L = _lossdict_to_np(losses)
    for k, vs in ref_losses.items():
        r = np.array([float(v) for v in vs])
        assert np.allclose(L[k], r, rtol=RTOL, atol=1e-7), (k, L[k], r)
    total, _ = parse_losses(losses)
    total.backward()
    params = dict(model.named_parameters())
    errs, num, den = [], 0.0, 0.0
    for k, v in sd.items():
        if not (O.trainable(k) and v.dtype == torch.float32):
            continue
        a, b = params[k].grad.cpu().double(), v.grad.double()
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
        if float(b.norm()) > 1e-12:
            errs.append(float((a - b).norm() / b.norm()))
            assert errs[-1] < 5e-2, (k, errs[-1])
    assert float(np.median(errs)) < 1e-3 and (num / den) ** 0.5 < 2e-2, (float(np.median(errs)), (num / den) ** 0.5)
    print("grad rel L2: median %.2e max %.2e global %.2e" % (float(np.median(errs)), max(errs), (num / den) ** 0.5))
    # The bounds above are against the fp32 CPU reference, which is itself one ReLU flip away from the exact gradients at this
    # size.  Anchor to an fp64 evaluation of the same step (as test_gpu_parity_full.py does at BASELINE size): this
    # implementation must be as close to fp64 as the reference's own fp32 arithmetic is.
    t64 = {k: (v.double() if v.is_floating_point() else v) for k, v in tsd.items()}
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in ssd.items()}
    s64 = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.is_floating_point() else v) for k, v in s64.items()}
    O.parse_losses(O.erd_step_loss(t64, s64, x.double(), boxes, labels, metas, 40, 80)).backward()

    def to64(g):
        errs, num, den = [], 0.0, 0.0
        for k, v in s64.items():
            if v.grad is None or float(v.grad.norm()) < 1e-12:
                continue
            a, b = g(k), v.grad
            errs.append(float((a - b).norm() / b.norm()))
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        return float(np.median(errs)), (num / den) ** 0.5, max(errs)

    hip64 = to64(lambda k: params[k].grad.cpu().double())
    cpu64 = to64(lambda k: sd[k].grad.double())
    print("rel L2 to fp64 (median / all elements / worst tensor): cpu fp32 %.2e %.2e %.2e | hip %.2e %.2e %.2e" % (cpu64 + hip64))
    assert hip64[1] <= max(1e-3, 1.5 * cpu64[1]) and hip64[0] <= max(1e-3, 1.5 * cpu64[0]) and hip64[2] <= max(1e-2, 1.5 * cpu64[2]), (hip64, cpu64)
    # ERS index sets end to end (teacher logits come from the HIP conv stack here)
    t_cls, t_bbox, sizes = model.ori_model._forward_cat(x.cuda())
    ers = model.sel_pos_cat(t_cls, t_bbox)
    cnt = ers["counts"].cpu()
    for i in range(3):
        assert torch.equal(ers["idx_cls"][i, :int(cnt[i, 0])].cpu(), aux["ers_cls"][i])
        assert torch.equal(ers["idx_bbox"][i, :int(cnt[i, 1])].cpu(), aux["ers_bbox"][i])


def test_plain_gfl_first40_loss_vs_oracle():
    """BASELINE configs[0]: gfl_r50_fpn first_40_cats forward + loss (no teacher)."""
    import erd_amd
    from erd_amd import Config, MODELS, parse_losses
    from e2e_util import CFG_FIRST
    tsd = O.procedural_state_dict(40, seed=0)
    model = MODELS.build(Config.fromfile(CFG_FIRST).model)
    model.load_state_dict(tsd, strict=True)
    model = model.cuda().train()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=3)
    x, metas = O.preprocess(imgs)
    losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
    cls, bbox = O.gfl_forward(tsd, x)
    ref = O.gfl_head_loss(cls, bbox, boxes, labels, metas, 40)
    L = _lossdict_to_np(losses)
    for k, vs in ref.items():
        assert np.allclose(L[k], np.array([float(v) for v in vs]), rtol=RTOL, atol=1e-7), k


def test_no_cpu_fallback():
    tsd, ssd = f7_state_dicts()
    import erd_amd
    from erd_amd import Config, MODELS
    from e2e_util import CFG_FIRST
    model = MODELS.build(Config.fromfile(CFG_FIRST).model)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 64, 64), mode="tensor")


def test_trainer_three_steps_follow_oracle_sgd_trajectory():
    """ERDTrainer (deferred SGD behind the next teacher forward, flat buffers, LinearLR warm-up, weight decay,
    momentum) against the oracle: 3 optimisation steps on two alternating batches; per-step loss and the parameter
    displacement after the last step (R22)."""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd)
    tr = ERDTrainer(model, lr=0.02, momentum=0.9, weight_decay=1e-4, batch_size_per_gpu=2, auto_scale_lr=False,
                    warmup_iters=3, warmup_start_factor=0.5)
    batches = []
    for s in (0, 1):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=s)
        x, metas = O.preprocess(imgs)
        batches.append((x, boxes, labels, metas))
    # oracle trajectory -- and the oracle's trajectory from weights perturbed by 1e-6 (relative): at 123x153 a handful of
    # ReLU / ATSS decisions sit on their thresholds, and how far two fp32 evaluations of the SAME mathematics drift apart
    # in three updates is a property of the problem, not of an implementation
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]

    def oracle_run(noise):
        sd = {k: v.clone() for k, v in ssd.items()}
        if noise:
            gen = torch.Generator().manual_seed(11)
            for k in names:
                sd[k] = sd[k] * (1 + noise * torch.randn(sd[k].shape, generator=gen))
        start = {k: sd[k].clone() for k in names}
        bufs, losses = {}, []
        for it in range(3):
            x, boxes, labels, metas = batches[it % 2]
            leaf = {k: (sd[k].clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
            total = O.parse_losses(O.erd_step_loss(tsd, leaf, x, boxes, labels, metas, 40, 80))
            total.backward()
            losses.append(float(total))
            lr = tr.lr_at(it)
            assert lr == pytest.approx(0.02 * (0.5 + 0.5 * it / 2))
            O.sgd_momentum_step({k: sd[k] for k in names}, {k: leaf[k].grad for k in names}, bufs, lr, 0.9, 1e-4)
        return sd, losses, {k: (sd[k] - start[k]).double() for k in names}

    sd, ref_loss, d_ref = oracle_run(0.0)
    _, _, d_eps = oracle_run(1e-6)
    den = sum(float(d_ref[k].pow(2).sum()) for k in names)
    self_err = (sum(float((d_eps[k] - d_ref[k]).pow(2).sum()) for k in names) / den) ** 0.5
    # Both fp32 forms of the direct launches, each with its OWN named bound: "f32" (every launch on the fp32 matrix
    # cores) keeps the original 3e-2; "f32x3" (three-limb products, the default) is held to the larger of 3e-2 and three times
    # `self_err`, the displacement error of the oracle's own trajectory from weights perturbed by 1e-6 -- at 123x153 a handful
    # of ReLU / ATSS decisions sit on their thresholds and which implementation owns a flip is chance (observed in round 3 with
    # truncating limbs: 1.6e-2 native, 5.2e-2 three-limb; the full-size, 48-seed statistics are profiles/r04_parity_seeds.json).
    # The per-step losses are the tight check (2e-3); the displacement catches a wrong lr / momentum / weight decay (a factor of
    # two in any of them reads > 0.3 here).
    from erd_amd import kernels as K
    errs = {}
    for mode in ("f32", K.DEFAULT_COMPUTE):
        K.set_compute(mode)
        try:
            model = build_erd(tsd, ssd)
            tr = ERDTrainer(model, lr=0.02, momentum=0.9, weight_decay=1e-4, batch_size_per_gpu=2, auto_scale_lr=False,
                            warmup_iters=3, warmup_start_factor=0.5)
            got = []
            for it in range(3):
                x, boxes, labels, metas = batches[it % 2]
                got.append(float(tr.train_step(x.cuda(), make_samples(boxes, labels, metas))["loss"]))
            tr.flush()
            torch.cuda.synchronize()
        finally:
            K.set_compute(K.DEFAULT_COMPUTE)
        assert np.allclose(got, ref_loss, rtol=2e-3), (mode, got, ref_loss)
        params = dict(model.named_parameters())
        num = sum(float(((params[k].detach().cpu() - ssd[k]).double() - d_ref[k]).pow(2).sum()) for k in names)
        errs[mode] = err = (num / den) ** 0.5
        print("%s: 3-step displacement rel L2 err: %.2e (the oracle's own, from weights x (1 + 1e-6 noise): %.2e); losses %s vs %s"
              % (mode, err, self_err, got, ref_loss))
        bound = 3e-2 if mode == "f32" else min(max(3e-2, 3.0 * self_err), 0.15)
        assert den > 0 and err < bound, (mode, err, bound, self_err)
    # frozen parts did not move; the teacher is untouched
    for k, v in ssd.items():
        if k not in names and v.dtype == torch.float32:
            assert torch.equal(dict(model.state_dict())[k].cpu(), v), k


def test_teacher_hipgraph_replay_equals_eager_on_mixed_resolutions():
    """BASELINE.json configs[4]: the frozen teacher's pass captured into a hipGraph per padded shape; replays on new
    pixels give bit-identical teacher logits, ERS index lists and NMS masks; a trainer using it logs the same losses."""
    from erd_amd.engine import ERDTrainer, TeacherGraphs
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd)
    tg = TeacherGraphs(model)
    side = torch.cuda.Stream()
    rng = np.random.RandomState(3)
    shapes = [(2, 3, 128, 160), (2, 3, 96, 224), (2, 3, 128, 160), (2, 3, 96, 224), (2, 3, 128, 160)]
    for i, shp in enumerate(shapes):
        x = torch.from_numpy(rng.standard_normal(shp).astype(np.float32)).cuda()
        with torch.no_grad():
            ref = model.teacher_pass(x)
        want = [t.clone() for t in ref.tensors()]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got = tg.run(x)
        torch.cuda.current_stream().wait_stream(side)
        assert got.sizes == ref.sizes
        for a, b in zip(got.tensors(), want):
            if a.dtype == torch.int64 and a.dim() == 2:       # ERS index lists: only the first count entries are defined
                continue
            assert torch.equal(a, b), (i, shp)
        cnt = got.ers["counts"].cpu()
        for n in range(shp[0]):
            for k, name in enumerate(("idx_cls", "idx_bbox")):
                assert torch.equal(got.ers[name][n, :int(cnt[n, k])], ref.ers[name][n, :int(cnt[n, k])])
    assert len(tg.graphs) == 2
    # trainer equivalence (two steps, alternating shapes)
    logs = {}
    for use_graph in (False, True):
        m = build_erd(tsd, ssd)
        tr = ERDTrainer(m, lr=0.01, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0, teacher_graph=use_graph)
        out = []
        for s in (0, 1, 0):
            imgs, boxes, labels = O.synthetic_batch(2, 123 if s == 0 else 90, 153 if s == 0 else 220, 40, seed=s)
            x, metas = O.preprocess(imgs)
            out.append(float(tr.train_step(x.cuda(), make_samples(boxes, labels, metas))["loss"].detach()))
        tr.flush()
        logs[use_graph] = out
    assert np.allclose(logs[True], logs[False], rtol=1e-5), logs


def test_teacher_hipgraph_with_look_ahead_and_shared_trunk():
    """The graph teacher under the benched schedule: the replay for step t+1 is queued next to the backward pass of step t,
    which still reads step t's teacher outputs and the shared trunk map out of the graph's static buffers -- two graphs per
    shape (buffer parity) keep them apart.  Same-shape and alternating-shape sequences log the eager trainer's losses, and
    the student reads the trunk the graph computed (no second trunk launch set)."""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()

    def batch(h, w, seed):
        imgs, boxes, labels = O.synthetic_batch(2, h, w, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        return x.cuda(), make_samples(boxes, labels, metas)

    for name, seq in (("same shape", [batch(123, 153, s) for s in range(4)]),
                      ("alternating", [batch(123, 153, 0), batch(90, 220, 1), batch(123, 153, 2), batch(90, 220, 3), batch(123, 153, 4)])):
        logs = {}
        for use_graph in (False, True):
            m = build_erd(tsd, ssd)
            tr = ERDTrainer(m, lr=0.01, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0, teacher_graph=use_graph)
            out = []
            for i, b in enumerate(seq):
                nxt = seq[i + 1] if i + 1 < len(seq) else None
                out.append(float(tr.train_step(*b, next_batch=nxt)["loss"].detach()))
                if use_graph and nxt is not None:
                    assert tr._teacher_ahead is not None and tr._teacher_ahead[1].trunk is not None      # queued ahead, trunk shared
            tr.flush()
            torch.cuda.synchronize()
            logs[use_graph] = out
            if use_graph:      # one graph per (padded shape, buffer parity) that occurred
                want = {(tuple(b[0].shape), i & 1) for i, b in enumerate(seq)}
                assert {(k[0], k[3]) for k in tr.teacher_graphs.graphs} == want, tr.teacher_graphs.graphs.keys()
        assert np.allclose(logs[True], logs[False], rtol=2e-4), (name, logs)
        assert logs[True][0] == pytest.approx(logs[False][0], rel=5e-6), (name, logs)      # step 0: identical weights


def test_bf16_matrix_core_mode_vs_oracle_bf16_multiplicands():
    """BASELINE.json configs[2]: every 1x1/3x3 convolution on the bf16 matrix cores (multiplicands rounded to bf16,
    exact products, fp32 accumulation / normalisation / losses / storage), against the oracle in its bf16-multiplicand
    mode.  A single convolution agrees to 2e-5 (test_gpu_kernels.py: exact products, only the summation order
    differs); through ~60 layers the two sides round activations that differ by 1e-7 to DIFFERENT bf16 neighbours now
    and then, each such flip is a 0.4 % perturbation that makes further flips likelier, and the difference settles at
    the bf16 quantisation-noise floor (tests/diag/debug_bf16.py: 2e-4 after layer1, 5e-3 from C4 on; 3e-7 in fp32 mode).
    Whole-network quantities therefore agree at bf16 resolution, not fp32: logits 0.1 abs (observed 0.03), losses
    2e-2 rel (distillation terms 1e-1), ERS index sets >= 75 % overlap.  Gradients at this tiny image size sit on a
    high noise floor in ANY bf16-multiplicand implementation: the oracle's own bf16-mode gradients move by 0.20 (median
    per-tensor rel L2; max 0.38) under a 2e-6 input perturbation, so the bounds here are 0.6 per tensor (>= 4096 elements), 0.3 median, 0.35 global.  (It must also differ measurably
    from the fp32 result -- the mode is really on.)"""
    from erd_amd import kernels as K, parse_losses
    tsd, ssd = f7_state_dicts()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
    x, metas = O.preprocess(imgs)
    sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v) for k, v in ssd.items()}
    with O.bf16_multiplicands():
        t_cls_ref, t_bbox_ref = O.gfl_forward(tsd, x)
        ref_losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
        O.parse_losses(ref_losses).backward()
    f32_losses = O.erd_step_loss(tsd, ssd, x, boxes, labels, metas, 40, 80)
    K.set_compute("bf16")
    try:
        model = build_erd(tsd, ssd)
        t_cls, t_bbox = model.ori_model(x.cuda(), mode="tensor")
        dmax = max(float((a.cpu() - b).abs().max()) for a, b in zip(t_cls, t_cls_ref))
        assert dmax < 0.1, dmax
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        L = _lossdict_to_np(losses)
        lerr = 0.0
        for k, vs in ref_losses.items():
            r = np.array([float(v) for v in vs])
            # the distillation terms are differences of two nearly equal (teacher / student) noisy responses
            assert np.allclose(L[k], r, rtol=1e-1 if k.startswith("loss_dist") else 2e-2, atol=1e-6), (k, L[k], r)
            lerr = max(lerr, float(np.max(np.abs(L[k] - r) / np.maximum(np.abs(r), 1e-6))))
        assert abs(float(sum(L["loss_cls"])) - float(sum(float(v) for v in f32_losses["loss_cls"]))) > 1e-6
        total, _ = parse_losses(losses)
        total.backward()
        tc, tb, sizes = model.ori_model._forward_cat(x.cuda())
        ers = model.sel_pos_cat(tc, tb)
        cnt = ers["counts"].cpu()
        jac = []
        for i in range(2):
            for name, key, col in (("idx_cls", "ers_cls", 0), ("idx_bbox", "ers_bbox", 1)):
                a = set(ers[name][i, :int(cnt[i, col])].cpu().tolist())
                b = set(aux[key][i].tolist())
                jac.append(len(a & b) / max(len(a | b), 1))
        assert min(jac) > 0.75, jac          # (sets of ~10 anchors at this image size: one borderline anchor = 0.1)
        params = dict(model.named_parameters())
        errs, num, den = [], 0.0, 0.0
        for k, v in sd.items():
            if not (O.trainable(k) and v.dtype == torch.float32) or float(v.grad.norm()) < 1e-12:
                continue
            a, b = params[k].grad.cpu().double(), v.grad.double()
            errs.append(float((a - b).norm() / b.norm()))
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            if b.numel() >= 4096:        # (scalars / short vectors are sums with heavy cancellation: no per-tensor bound)
                assert errs[-1] < 0.6, (k, errs[-1])
        assert float(np.median(errs)) < 0.3 and (num / den) ** 0.5 < 0.35, (float(np.median(errs)), (num / den) ** 0.5)
        print("bf16 mode vs oracle(bf16 multiplicands): logits max abs %.2e, losses max rel %.2e, ERS Jaccard min %.3f, "
              "grad rel L2 median %.2e max %.2e" % (dmax, lerr, min(jac), float(np.median(errs)), max(errs)))
    finally:
        K.set_compute(K.DEFAULT_COMPUTE)


def test_teacher_look_ahead_follows_the_plain_trainer():
    """`train_step(..., next_batch=...)`: the frozen teacher's half of step t+1 is queued next to the backward pass of step t
    (it depends on nothing step t updates).  (a) What is queued ahead is what an in-place teacher pass on that batch returns
    (logits, ERS lists, NMS mask, ATSS targets), checked while the backward it overlapped is still fresh; (b) three steps on
    alternating batches log the losses of the plain order (float-atomic-order noise, amplified by two updates at lr 0.02:
    observed up to 2e-5); (c) a step whose `inputs` is not the announced tensor recomputes its teacher."""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    batches = []
    for seed in (0, 1):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        batches.append((x.cuda(), make_samples(boxes, labels, metas)))

    def run(ahead):
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
        logs = []
        for i in range(3):
            lv = tr.train_step(*batches[i % 2], next_batch=batches[(i + 1) % 2] if ahead else None)
            logs.append({k: float(v) for k, v in lv.items()})
            if ahead:                                                        # (a)
                inp, t = tr._teacher_ahead
                assert inp is batches[(i + 1) % 2][0]
                torch.cuda.synchronize()
                with torch.no_grad():
                    ref = model.teacher_pass(*batches[(i + 1) % 2])
                assert torch.equal(t.t_cls, ref.t_cls) and torch.equal(t.t_bbox, ref.t_bbox) and torch.equal(t.keep, ref.keep)
                assert torch.equal(t.ers["counts"], ref.ers["counts"])
                cnt = ref.ers["counts"].cpu()
                for n in range(cnt.shape[0]):                                # (entries past the counts are unspecified)
                    for col, k in enumerate(("idx_cls", "idx_bbox")):
                        assert torch.equal(t.ers[k][n, :int(cnt[n, col])], ref.ers[k][n, :int(cnt[n, col])]), (n, k)
                assert torch.equal(t.targets.labels, ref.targets.labels) and torch.equal(t.targets.bbox_targets, ref.targets.bbox_targets)
        if ahead:                                                            # (c)
            stale = tr._teacher_ahead[1]
            lv = tr.train_step(batches[1][0].clone(), batches[1][1])
            assert tr._teacher_ahead is None and torch.isfinite(lv["loss"]).item() and stale is not None
        tr.flush()
        torch.cuda.synchronize()
        return logs

    plain_logs = run(False)
    ahead_logs = run(True)
    for i, (a, b) in enumerate(zip(plain_logs, ahead_logs)):                 # (b)
        for k in a:
            assert b[k] == pytest.approx(a[k], rel=5e-6 if i == 0 else 2e-4, abs=1e-6), (i, k, a[k], b[k])


def test_teacher_look_ahead_waits_for_the_producer_of_the_next_batch():
    """Runner.train prepares batch t+1 on the CURRENT stream (preprocess / resize kernels, non-blocking H2D copies, no
    host sync) before it calls train_step(t, next_batch=t+1).  The look-ahead teacher runs on the side stream: it must be
    ordered behind that producer.  Here the next batch's pixels are written by a copy that sits behind a long-running
    kernel on the current stream; without the join the side stream (idle after its first step) would read the
    placeholder values.  The queued teacher output must equal an in-place teacher pass on the FINAL pixels."""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    batches = []
    for seed in (0, 1, 2):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        batches.append((x.cuda(), make_samples(boxes, labels, metas)))
    model = build_erd(tsd, ssd)
    tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
    tr.train_step(*batches[0], next_batch=batches[1])          # the side stream has run once: look-ahead is active from here
    torch.cuda.synchronize()
    for rep in range(3):
        final = batches[2][0] + float(rep)
        slot = torch.full_like(final, 1e3)                     # placeholder pixels the teacher must never see
        nxt = (slot, batches[2][1])
        torch.cuda._sleep(400_000_000)                         # ~0.2 s of spinning on the current stream ...
        slot.copy_(final)                                      # ... and only then the producer of the next batch
        tr.train_step(*batches[1], next_batch=nxt)
        inp, t = tr._teacher_ahead
        assert inp is slot
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = model.teacher_pass(final, batches[2][1])
        assert torch.equal(t.t_cls, ref.t_cls) and torch.equal(t.t_bbox, ref.t_bbox), rep
        assert torch.equal(t.ers["counts"], ref.ers["counts"]) and torch.equal(t.keep, ref.keep)
        tr._teacher_ahead = None                               # (the announced tensor is not fed back in this test)
    tr.flush()
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_per_bucket_update_equals_the_one_launch_update(mode, monkeypatch):
    """ERDTrainer updates per gradient bucket while the rest of the backward pass runs (engine.BucketedGradSync on_bucket: SGD on the
    bucket's slice, its BN folds, its prepared weights, on a side stream; only the tail bucket is left at the step boundary).  After
    every step the flat parameters and momenta are BIT-equal to ONE erd_sgd_momentum launch over the whole buffers applied to the
    state before the step and the gradient the step left in the flat buffer; all buckets but the tail were released inside
    backward; the look-ahead teacher and the warm-up schedule (a new learning rate every step) go through the same path."""
    from erd_amd.engine import ERDTrainer
    from erd_amd import kernels as K
    tsd, ssd = f7_state_dicts()
    batches = []
    for seed in (0, 1):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        batches.append((x.cuda(), make_samples(boxes, labels, metas)))
    monkeypatch.setenv("ERD_BUCKET_UPDATE", "1")      # (the bf16 mode's default is the update at the step boundary since round 6)
    K.set_compute(mode)
    try:
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=5, bucket_mb=1)
        assert tr.bucket_update and tr.sync is not None and not tr.sync.reduce
        nb = len(tr.flat.buckets)
        assert nb >= 4
        for s, e, mem in tr.flat.buckets:                 # whole blocks per bucket
            keys = {tr._block_of(tr.flat.names[i]) for i in mem}
            assert all((tr._block_of(n) in keys) == (i in mem) for i, n in enumerate(tr.flat.names))
        for i in range(4):
            d0, m0 = tr.flat.data.clone(), tr.flat.momentum.clone()
            first = tr._first
            tr.train_step(*batches[i % 2], next_batch=batches[(i + 1) % 2])
            assert tr.sync.issued_in_backward == nb - 1 and tr.sync.late_buckets == 0, (tr.sync.issued_in_backward, nb, tr.sync.missing)
            tr.flush()
            torch.cuda.synchronize()
            assert tr.last_lr == pytest.approx(tr.lr_at(i))
            K.sgd_momentum_(d0, tr.flat.grad.clone(), m0, tr.last_lr, tr.momentum, tr.weight_decay, 1.0, first)
            assert torch.equal(d0, tr.flat.data) and torch.equal(m0, tr.flat.momentum), i
            if mode == "bf16":
                assert torch.equal(tr.flat.data_bf16, tr.flat.data.to(torch.bfloat16))
        assert not tr._first
    finally:
        K.set_compute(K.DEFAULT_COMPUTE)


def test_batched_bn_fold_equals_the_per_layer_fold_and_tracks_updates():
    """ERDTrainer folds every trainable frozen-statistics BN of the student in one launch after each optimizer update
    (functional.BnPrefold, erd_bn_fold_batch): the views it hands to the forward pass are bit-equal to a per-layer erd_bn_fold of
    the CURRENT parameters, step after step; parameters changed behind the trainer's back fall back to the per-layer launch."""
    from erd_amd.engine import ERDTrainer
    from erd_amd import functional as Fn, kernels as K
    tsd, ssd = f7_state_dicts()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
    x, metas = O.preprocess(imgs)
    batch = (x.cuda(), make_samples(boxes, labels, metas))
    model = build_erd(tsd, ssd)
    tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
    assert tr.prefold is not None and len(tr.prefold.bns) > 30
    for step in range(3):
        tr.train_step(*batch)
        tr.flush()                                  # the update + the batched fold of the updated parameters
        assert tr.prefold.valid[0]
        for m in tr.prefold.bns:
            sc, sh = Fn._bn_fold_cached(m.weight, m.bias, m.running_mean, m.running_var, m.eps)
            rs, rh = K.bn_fold(m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var, m.eps)
            assert sc.data_ptr() == m.weight._erd_prefold[2].data_ptr()            # served from the batched result
            assert torch.equal(sc, rs) and torch.equal(sh, rh)
    m = tr.prefold.bns[0]
    with torch.no_grad():
        m.weight.mul_(1.5)                          # (bumps the version counter: the entry no longer matches)
    sc, _ = Fn._bn_fold_cached(m.weight, m.bias, m.running_mean, m.running_var, m.eps)
    assert sc.data_ptr() != m.weight._erd_prefold[2].data_ptr()
    assert torch.equal(sc, K.bn_fold(m.weight.detach(), m.bias.detach(), m.running_mean, m.running_var, m.eps)[0])


@pytest.mark.parametrize("mode", ["f32", "f32x3", "bf16"])
def test_prepared_weight_buffers_equal_the_inline_transforms(mode):
    """kernels.ParamPrep: after every optimizer update the trainer rebuilds the transposed (BN-scaled) weights of the
    input-gradient convolutions and the Winograd weight images in two launches (erd_weight_prep_batch).  From the step at
    which a recipe is known on, the wrappers serve those buffers -- and they must be bit-equal to what erd_weight_transpose
    / erd_wino_weights produce from the CURRENT parameters."""
    from erd_amd.engine import ERDTrainer
    from erd_amd import functional as Fn, kernels as K
    tsd, ssd = f7_state_dicts()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
    x, metas = O.preprocess(imgs)
    batch = (x.cuda(), make_samples(boxes, labels, metas))
    K.set_compute(mode)
    try:
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
        assert tr.prep is not None and all(p._erd_prep is tr.prep for p in tr.flat.params)
        for _ in range(4):
            tr.train_step(*batch)
        tr.flush()
        torch.cuda.synchronize()
        kinds = {}
        for key, r in tr.prep.recipes.items():
            assert r.stamp == tr.prep.stamp, key                      # rebuilt by the last update
            kinds[key[0]] = kinds.get(key[0], 0) + 1
            if r.kind == 2:
                ref = torch.empty_like(r.out)
                K.call("erd_wino_weights", K._p(r.src), K._p(ref), r.Cout, r.Cin, r.flip, K._stream())
            elif r.kind == 4:                     # the Winograd weight image in the three-limb layout ("f32x3")
                ref = torch.empty_like(r.out)
                K.call("erd_wino_weights_x3", K._p(r.src), K._p(ref), r.Cout, r.Cin, r.flip, K._stream())
            elif r.kind == 3:                     # the three bf16 limb planes of a weight ("f32x3")
                ref = torch.empty_like(r.out)
                K.call("erd_split3", K._p(r.src), K._p(ref), r.src.numel(), K._stream())
                assert torch.equal((r.out[0].float() + r.out[1].float()) + r.out[2].float(), r.src), key      # limbs sum to the weight
            else:
                ref = torch.empty_like(r.out)
                K.call("erd_weight_transpose_bf16" if r.kind == 1 else "erd_weight_transpose", K._p(r.src), K._p(r.rowscale), K._p(ref),
                       r.Cout, r.ntaps, r.Cin, r.flip, K._stream())
            assert torch.equal(ref, r.out), key
        assert kinds.get("T", 0) >= 40, kinds                         # every trainable convolution's transposed weights
        if mode == "f32":
            assert kinds.get("U", 0) >= 15 and kinds.get("UT", 0) >= 15, kinds
        if mode == "f32x3":                                           # (the Winograd launches run in the three-limb form too)
            assert kinds.get("U3", 0) >= 15 and kinds.get("UT3", 0) >= 15, kinds
        if mode == "f32x3":
            assert kinds.get("X", 0) >= 30 and kinds.get("XT", 0) >= 25, kinds    # the direct launches' limb planes, both forms
        # ... and they are what the wrappers hand out
        blk = model.backbone.layer3[0]
        wk = Fn.ohwi(blk.conv2.weight)
        r = tr.prep.recipes[("T", id(blk.conv2.weight), mode == "bf16")]
        assert K.weight_transpose(wk, r.rowscale).data_ptr() == r.out.data_ptr()
        with torch.no_grad():
            blk.conv2.weight.mul_(1.0)              # version bump: the prepared buffer no longer vouches for the parameter
        assert K.weight_transpose(wk, r.rowscale).data_ptr() != r.out.data_ptr()
    finally:
        K.set_compute(K.DEFAULT_COMPUTE)


def test_bf16_full_size_step_against_the_fp32_path():
    """BASELINE.json configs[2] at BASELINE size (800x1333, one image): the bf16 mode -- bf16 matrix cores, feature maps and
    their gradients STORED as bf16, fp32 statistics / head outputs / losses / parameter gradients -- against this package's
    fp32 path (itself pinned to the oracle by test_gpu_parity_full.py) on the same weights and batch.  At this size every
    gradient is a sum over 10^4..10^5 pixels, so bf16 noise averages out and the comparison can see a real defect (a dropped
    term, a wrong scale, a stale buffer) that the tiny-image bounds above cannot.  Measured (tests/diag/diag_bf16_fullsize.py,
    seeds 7 / 8): total loss 1e-5 / 5e-4 relative, worst single loss 7 % (loss_dist_bbox: a difference of two nearly equal noisy
    responses), whole gradient cosine 0.991 / 0.992 with norm ratio 1.004 / 0.995, per-tensor cosine >= 0.93, ERS index sets
    0.93 Jaccard (~750 anchors each); the round-1 form (bf16 multiplicands, fp32 maps) measures the same on every line."""
    from erd_amd import kernels as K, parse_losses
    tsd, ssd = f7_state_dicts()
    imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=7)
    x, metas = O.preprocess(imgs)

    def run(mode):
        K.set_compute(mode)
        try:
            model = build_erd(tsd, ssd)
            losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
            total, lv = parse_losses(losses)
            total.backward()
            tc, tb, _ = model.ori_model._forward_cat(x.cuda())
            ers = model.sel_pos_cat(tc, tb)
            cnt = ers["counts"].cpu()
            sets = [set(ers[n][0, :int(cnt[0, c])].cpu().tolist()) for n, c in (("idx_cls", 0), ("idx_bbox", 1))]
            feats = model.extract_feat(x.cuda()) if hasattr(model, "extract_feat") else None
            g = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
            return {k: float(v) for k, v in lv.items()}, g, sets, (None if feats is None else feats[0].dtype)
        finally:
            K.set_compute(K.DEFAULT_COMPUTE)

    l32, g32, s32, _ = run("f32")
    l16, g16, s16, fdt = run("bf16")
    assert fdt in (None, torch.bfloat16), fdt                  # the maps really are stored as bf16
    assert abs(l16["loss"] - l32["loss"]) <= 2e-3 * abs(l32["loss"]), (l16["loss"], l32["loss"])
    assert l16["loss"] != l32["loss"]                          # ... and the mode really is on
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 0.12 * abs(l32[k]) + 1e-6, (k, l16[k], l32[k])
    dot = na = nb = 0.0
    for k, b in g32.items():
        a = g16[k]
        if float(b.norm()) < 1e-12:
            continue
        c = float((a * b).sum() / (a.norm() * b.norm()))
        if b.numel() >= 4096:
            assert c >= 0.9, (k, c)
        dot += float((a * b).sum()); na += float(a.pow(2).sum()); nb += float(b.pow(2).sum())
    cos, ratio = dot / (na * nb) ** 0.5, (na / nb) ** 0.5
    assert cos >= 0.985 and 0.98 <= ratio <= 1.02, (cos, ratio)
    jac = [len(a & b) / max(len(a | b), 1) for a, b in zip(s16, s32)]
    assert min(jac) >= 0.88, jac
    print("bf16 (bf16-stored maps) vs fp32 at 800x1333: total loss rel %.2e, gradient cosine %.4f, norm ratio %.4f, ERS Jaccard %s"
          % (abs(l16["loss"] - l32["loss"]) / abs(l32["loss"]), cos, ratio, ["%.3f" % j for j in jac]))


def test_bf16_full_size_step_against_the_oracle_bf16_mode():
    """The same bf16 step against the ORACLE in its bf16 mode (`O.bf16_multiplicands()`: both multiplicands of every 1x1 / 3x3
    convolution rounded to bf16, fp32 accumulation -- what the reference's AMP switch does to the convolutions, tools/train.py:85-97)
    at BASELINE size, not against this package's own fp32 path.  END TO END this can only be a SANITY bound at the bf16 noise floor:
    a network whose maps are rounded to bf16 is chaotic at the 4e-3 level (a 1e-7 difference in summation order lands on the other
    side of a rounding boundary now and then, every flip is a full bf16 ulp, and flips multiply from layer to layer), so that ANY two
    bf16 implementations -- this path against the oracle's `bf16_stored_maps` mode with identical rounding points included
    (tools/dbg/bf16_parity_probe.py: 1 - cos 0.0092 against 0.0118 for `bf16_multiplicands`, the floor 0.0104-0.0109) -- are as far
    from each other as each is from fp32, and a 1 % defect in one kernel does not move these numbers (measured 0.98824
    against 0.98956).  The bound that CAN fail on such a defect is unit-wise and teacher-forced: tests/test_gpu_bf16_stagewise.py.
    Asserted here, relative to the measured floor: direction, norm, losses and ERS overlap no worse than the oracle's own bf16-vs-fp32."""
    from erd_amd import kernels as K, parse_losses
    tsd, ssd = f7_state_dicts()
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]
    imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=7)
    x, metas = O.preprocess(imgs)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))

    def oracle(bf16):
        sd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in ssd.items()}
        if bf16:
            with O.bf16_multiplicands():
                losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
                O.parse_losses(losses).backward()
        else:
            losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
            O.parse_losses(losses).backward()
        row = {k: float(sum(v.detach().mean() for v in vs)) for k, vs in losses.items()}
        row["loss"] = float(O.parse_losses(losses).detach())
        return row, {k: sd[k].grad.double() for k in names}, [set(aux["ers_cls"][0].tolist()), set(aux["ers_bbox"][0].tolist())]

    try:
        l_o16, g_o16, s_o16 = oracle(True)
        l_o32, g_o32, s_o32 = oracle(False)
    finally:
        torch.set_num_threads(threads)
    K.set_compute("bf16")
    try:
        model = build_erd(tsd, ssd)
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        total, lv = parse_losses(losses)
        total.backward()
        t = model.teacher_pass(x.cuda())
        cnt = t.ers["counts"].cpu()
        s_h16 = [set(t.ers[n][0, :int(cnt[0, c])].cpu().tolist()) for n, c in (("idx_cls", 0), ("idx_bbox", 1))]
        p = dict(model.named_parameters())
        g_h16 = {k: p[k].grad.detach().cpu().double() for k in names}
        l_h16 = {k: float(v) for k, v in lv.items()}
    finally:
        K.set_compute(K.DEFAULT_COMPUTE)

    def cos(ga, gb):
        dot = na = nb = 0.0
        worst = 1.0
        for k in names:
            a, b = ga[k], gb[k]
            if float(b.norm()) < 1e-12:
                continue
            if b.numel() >= 4096:
                worst = min(worst, float((a * b).sum() / (a.norm() * b.norm())))
            dot += float((a * b).sum()); na += float(a.pow(2).sum()); nb += float(b.pow(2).sum())
        return dot / (na * nb) ** 0.5, (na / nb) ** 0.5, worst

    c_ho, r_ho, w_ho = cos(g_h16, g_o16)          # HIP bf16 against the oracle's bf16 mode
    c_oo, r_oo, w_oo = cos(g_o16, g_o32)          # the oracle's bf16 mode against the oracle's fp32: the rounding noise itself
    jac_ho = [len(a & b) / max(len(a | b), 1) for a, b in zip(s_h16, s_o16)]
    jac_oo = [len(a & b) / max(len(a | b), 1) for a, b in zip(s_o16, s_o32)]
    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-7)
    print("bf16 at 800x1333, HIP vs oracle-bf16 | oracle-bf16 vs oracle-fp32: gradient cosine %.5f | %.5f, norm ratio %.4f | %.4f, worst big tensor "
          "%.4f | %.4f, total loss rel %.2e | %.2e, worst loss entry %.2e | %.2e, ERS Jaccard %s | %s"
          % (c_ho, c_oo, r_ho, r_oo, w_ho, w_oo, rel(l_h16["loss"], l_o16["loss"]), rel(l_o16["loss"], l_o32["loss"]),
             max(rel(l_h16[k], v) for k, v in l_o16.items()), max(rel(l_o16[k], v) for k, v in l_o32.items()),
             ["%.3f" % j for j in jac_ho], ["%.3f" % j for j in jac_oo]))
    # the HIP bf16 step is closer to the oracle's bf16 mode than that mode is to fp32 -- in direction, norm and losses
    assert c_ho >= 0.985 and (1.0 - c_ho) <= 1.25 * (1.0 - c_oo) + 1e-4, (c_ho, c_oo)
    assert abs(r_ho - 1.0) <= max(0.02, 1.5 * abs(r_oo - 1.0)), (r_ho, r_oo)
    assert rel(l_h16["loss"], l_o16["loss"]) <= max(2e-3, 2.0 * rel(l_o16["loss"], l_o32["loss"])), (l_h16["loss"], l_o16["loss"])
    assert min(jac_ho) >= min(0.88, min(jac_oo) - 0.03), (jac_ho, jac_oo)


def test_shared_frozen_trunk_feeds_both_networks(monkeypatch):
    """Student stem + layer1 are frozen and warm-started from the teacher checkpoint (gfl_increment_erd.py:83-93,
    resnet.py:613-629): when the two copies are bit-identical the trunk is computed once per step and fed to both
    networks.  Same losses and gradients as two separate trunks; any difference between the copies turns sharing off."""
    from erd_amd import parse_losses
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd)
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=3)
    x, metas = O.preprocess(imgs)
    samples = make_samples(boxes, labels, metas)

    def step():
        model.zero_grad(set_to_none=True)
        total, log = parse_losses(model(x.cuda(), samples, mode="loss"))
        total.backward()
        return float(total), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    assert model.shares_trunk()
    with torch.no_grad():
        out = model.teacher_pass(x.cuda())
    assert out.trunk is not None and out.trunk[0].shape[-1] == 256          # C2, NHWC
    shared_total, shared_grads = step()
    monkeypatch.setenv("ERD_SHARE_TRUNK", "0")
    assert not model.shares_trunk()
    own_total, own_grads = step()
    assert shared_total == pytest.approx(own_total, rel=1e-5)
    num = sum(float((shared_grads[k] - own_grads[k]).double().pow(2).sum()) for k in own_grads)
    den = sum(float(own_grads[k].double().pow(2).sum()) for k in own_grads)
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5      # (the teacher's trunk moves from Winograd to the direct kernels)
    monkeypatch.delenv("ERD_SHARE_TRUNK")
    assert model.shares_trunk()
    with torch.no_grad():
        model.backbone.layer1[1].bn2.bias.add_(1e-3)              # the copies now differ: every network computes its own
    assert not model.shares_trunk()
    with torch.no_grad():
        assert model.teacher_pass(x.cuda()).trunk is None


def test_whole_step_hipgraph_replay_follows_the_eager_trainer():
    """`ERDTrainer(step_graph=True)`: everything between two SGD updates (teacher, ERS, NMS, ATSS targets, student forward,
    losses, backward on all streams) is captured once per input shape and replayed.  Same logged losses and the same
    weights after four steps on alternating batches as the eager trainer (the captured kernels are the eager ones)."""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    batches = []
    for seed in (0, 5):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        batches.append((x.cuda(), make_samples(boxes, labels, metas)))

    def run(step_graph):
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0, step_graph=step_graph)
        logs = []
        for i in range(4):
            lv = tr.train_step(*batches[i % 2])
            logs.append({k: float(v) for k, v in lv.items()})
        tr.flush()
        torch.cuda.synchronize()
        return logs, {k: v.detach().clone() for k, v in model.state_dict().items() if v.dtype == torch.float32}, tr

    eager_logs, eager_w, _ = run(False)
    eager2_logs, _, _ = run(False)
    noise = max(abs(a[k] - b[k]) / max(abs(a[k]), 1e-6) for a, b in zip(eager_logs, eager2_logs) for k in a)
    print("eager vs eager (float-atomic order only): max relative loss difference %.2e" % noise)
    graph_logs, graph_w, tr = run(True)
    assert tr.step_graph and len(tr._step_graphs) == 1
    for i, (a, b) in enumerate(zip(eager_logs, graph_logs)):
        assert a.keys() == b.keys()
        for k in a:
            # Step 0 runs on identical weights: the captured kernels are the eager ones, only the order of the float atomics
            # (column sums, GroupNorm statistics) differs -- the same noise two eager runs show after four steps (measured
            # 1e-6 .. 2e-5 relative from run to run).
            # From step 1 on the weights carry that noise through ReLU / top-k decisions: observed up to 2e-5 after four steps.
            tol = 5e-6 if i == 0 else 2e-4
            assert b[k] == pytest.approx(a[k], rel=tol, abs=1e-6), (i, k, a[k], b[k])
    num = sum(float((graph_w[k] - eager_w[k]).double().pow(2).sum()) for k in eager_w)
    den = sum(float(eager_w[k].double().pow(2).sum()) for k in eager_w)
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5


def test_step_graphs_of_two_shapes_survive_workspace_growth():
    """A captured graph holds the ADDRESSES of the workspaces its kernels used (split-K slabs on the trailing stream,
    GroupNorm statistics, ERS / NMS scratch).  Capturing a LARGER shape afterwards grows those workspaces; the outgrown
    buffers must stay allocated (kernels.pin_workspaces), otherwise the replay of the small graph writes into memory the
    caching allocator has handed to other tensors.  small -> large -> small replays: the small shape's logged losses are
    the eager trainer's, and a canary allocated after the growth is untouched."""
    from erd_amd.engine import ERDTrainer
    from erd_amd import kernels as K
    tsd, ssd = f7_state_dicts()

    def batch(h, w, seed):
        imgs, boxes, labels = O.synthetic_batch(2, h, w, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        return x.cuda(), make_samples(boxes, labels, metas)

    small, large = batch(96, 128, 0), batch(250, 330, 1)
    order = [small, large, small, large, small]

    def run(step_graph):
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.0, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0, step_graph=step_graph)
        logs = []
        canaries = []
        for b in order:
            lv = tr.train_step(*b)
            logs.append({k: float(v) for k, v in lv.items()})
            torch.cuda.synchronize()
            if step_graph:       # grab whatever the allocator has free right now (a freed workspace would be first in line)
                canaries.append(torch.zeros(1 << 20, device="cuda"))
        tr.flush()
        torch.cuda.synchronize()
        return logs, tr, canaries

    eager, _, _ = run(False)
    retired_before = len(K._WS_RETIRED)
    graph, tr, canaries = run(True)
    assert len(tr._step_graphs) == 2
    assert len(K._WS_RETIRED) > retired_before, "the larger shape must have outgrown at least one workspace"
    for c in canaries:
        assert float(c.abs().max()) == 0.0, "a replayed graph wrote into memory that was handed out again"
    for i, (a, b) in enumerate(zip(eager, graph)):      # lr = 0: every visit of a shape sees the same weights
        for k in a:
            assert b[k] == pytest.approx(a[k], rel=2e-5, abs=1e-6), (i, k, a[k], b[k])
