"""GPU, BASELINE.json full size (800x1344 padded, 22 400 anchors/image): size-independent properties of the step.
  * the stream-K work decomposition (persistent grid, cross-workgroup fix-up) and plain one-tile-per-workgroup
    launches give the same losses / gradients over several optimisation steps (sequences of launches with
    different tile counts share one workspace -- this is the regression test for that);
  * two identical runs are bit-identical in the forward losses (deterministic reductions);
  * ERS selects 1-8 % of the anchors, NMS keeps a non-empty subset, every parameter stays finite."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("oracle_threads")]      # (32 host threads for the oracle: tests/conftest.py)

from e2e_util import build_erd, f7_state_dicts


def _run(streamk: bool, steps: int = 3, bs: int = 2):
    import bench
    from erd_amd import kernels as K
    from erd_amd.engine import ERDTrainer
    old = K.STREAMK
    K.STREAMK = streamk
    try:
        tsd, ssd = f7_state_dicts()
        model = build_erd(tsd, ssd)
        tr = ERDTrainer(model, lr=0.01, batch_size_per_gpu=bs, auto_scale_lr=False, warmup_iters=0)
        batches = [bench.synthetic_gpu_batch(bs, seed=10 + i, device=torch.device("cuda", 0)) for i in range(2)]
        logs = []
        for i in range(steps):
            log = tr.train_step(*batches[i % 2])
            logs.append({k: float(v.detach()) for k, v in log.items()})
        tr.flush()
        torch.cuda.synchronize()
        t = model.teacher_pass(batches[0][0])
        return logs, model, t
    finally:
        K.STREAMK = old


def test_streamk_equals_tile_parallel_and_step_is_sane():
    logs_a, model_a, t = _run(True)
    logs_b, model_b, _ = _run(False)
    logs_c, _, _ = _run(True)
    for i, (a, b) in enumerate(zip(logs_a, logs_b)):
        for k in a:      # step 0 = pure forward (summation order differs only); later steps pass through SGD updates
            assert a[k] == pytest.approx(b[k], rel=1e-4 if i == 0 else 5e-3, abs=1e-6), (i, k, a[k], b[k])
    assert logs_a[0] == logs_c[0]                                   # run-to-run bit-identical first step
    pa, pb = dict(model_a.named_parameters()), dict(model_b.named_parameters())
    for k in ("bbox_head.gfl_cls.weight", "backbone.layer2.0.conv1.weight", "neck.fpn_convs.0.conv.weight"):
        assert torch.isfinite(pa[k]).all()
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-3 * float(pb[k].abs().max())
    A = t.t_cls.shape[1]
    assert A == 22400
    cnt = t.ers["counts"].cpu()
    assert ((cnt > 0.005 * A) & (cnt < 0.12 * A)).all(), cnt
    kc = t.keep_count.cpu()
    assert (kc > 0).all() and (kc <= cnt[:, 1]).all()
    assert 1.0 < logs_a[0]["loss"] < 10.0


def test_r101_70_plus_10_full_size_step_is_sane():
    """BASELINE.json configs[3] at full size: GFL-R101, 70 old + 10 new classes, 2 images of 800x1344."""
    import bench
    import e2e_util as U
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts(70, 80, 101)
    model = build_erd(tsd, ssd, cfg_first=U.CFG_FIRST70, cfg_incre=U.CFG_INCRE10)
    tr = ERDTrainer(model, lr=0.01, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
    batches = [bench.synthetic_gpu_batch(2, seed=20 + i, device=torch.device("cuda", 0), num_new=10) for i in range(2)]
    logs = [{k: float(v.detach()) for k, v in tr.train_step(*batches[i % 2]).items()} for i in range(3)]
    tr.flush()
    torch.cuda.synchronize()
    assert all(0.5 < l["loss"] < 20.0 for l in logs), logs
    assert len([k for k in logs[0] if k.startswith("loss_dist")]) == 2
    t = model.teacher_pass(batches[0][0])
    assert t.t_cls.shape == (2, 22400, 70)
    cnt = t.ers["counts"].cpu()
    assert ((cnt > 0.001 * 22400) & (cnt < 0.12 * 22400)).all(), cnt
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_bf16_mode_full_size_tracks_fp32():
    """BASELINE.json configs[2] at full size: three optimisation steps on the bf16 matrix cores stay within 2 % of the
    fp32 path's losses step by step (same batches, same initial weights), and every parameter stays finite."""
    from erd_amd import kernels as K
    logs = {}
    for mode in ("f32", "bf16"):
        K.set_compute(mode)
        try:
            logs[mode], model, _ = _run(True, steps=3, bs=2)
            assert all(torch.isfinite(p).all() for p in model.parameters())
        finally:
            K.set_compute(K.DEFAULT_COMPUTE)
    for a, b in zip(logs["bf16"], logs["f32"]):
        assert a["loss"] == pytest.approx(b["loss"], rel=2e-2), (a, b)
        assert a["loss"] != b["loss"]


@pytest.mark.parametrize("hw", [(800, 1216), (768, 1333), (704, 1088)])
def test_mixed_resolution_winograd_equals_direct(hw):
    """BASELINE.json configs[4] resolutions (padded to /32: level sizes with odd heights/widths, partial Winograd tiles on
    every level): the first optimisation step with the Winograd kernels reproduces the direct implicit-GEMM path."""
    import bench
    from erd_amd import kernels as K
    from erd_amd.engine import ERDTrainer
    H, W = hw
    old_hw = (bench.H, bench.W)
    logs = {}
    try:
        bench.H, bench.W = H, W
        for wino in (True, False):
            K.WINOGRAD = wino
            tsd, ssd = f7_state_dicts()
            model = build_erd(tsd, ssd)
            tr = ERDTrainer(model, lr=0.01, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
            batch = bench.synthetic_gpu_batch(2, seed=30, device=torch.device("cuda", 0))
            out = []
            for _ in range(2):
                log = tr.train_step(*batch)
                out.append({k: float(v.detach()) for k, v in log.items()})
            tr.flush()
            torch.cuda.synchronize()
            logs[wino] = out
            assert all(torch.isfinite(p).all() for p in model.parameters())
    finally:
        K.WINOGRAD = True
        bench.H, bench.W = old_hw
    # two valid fp32 evaluations of the same network: 1e-6-level differences in the teacher logits may move ONE anchor across
    # an ERS threshold (700-900 selected per image), which shows up as a few 1e-4 in the distillation terms; the second
    # step additionally carries the first update
    for i, (a, b) in enumerate(zip(logs[True], logs[False])):
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=(1e-3 if k.startswith("loss_dist") else 2e-4) * (1 + 2 * i), abs=1e-6), \
                (i, k, a[k], b[k])


def test_teacher_hipgraph_replay_at_the_full_mixed_resolutions():
    """BASELINE.json configs[4] in one piece: the captured teacher (`engine.TeacherGraphs`, one hipGraph per padded shape) REPLAYED at
    the three full-size mixed resolutions 800x1216 / 768x1344 / 704x1088 on new pixels each time -- teacher logits, ERS counts, index lists
    and NMS masks bit-identical to the eager teacher pass, and on one of the shapes the ERS index sets of the replay equal the CPU
    oracle's (the small-size twin of this test is tests/test_gpu_e2e.py::test_teacher_hipgraph_replay_equals_eager_on_mixed_resolutions)."""
    from erd_amd.engine import TeacherGraphs
    from oracle import erd_oracle as O
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd).eval()
    tg = TeacherGraphs(model)
    side = torch.cuda.Stream()
    g = torch.Generator().manual_seed(61)
    shapes = [(800, 1216), (768, 1344), (704, 1088)]
    for rep in range(2):            # the second visit of a shape is a pure replay of the graph captured on the first
        for hw in shapes:
            x = torch.randn((2, 3) + hw, generator=g).cuda()
            with torch.no_grad():
                ref = model.teacher_pass(x)
            want = [t.clone() for t in ref.tensors()]
            want_idx = {k: ref.ers[k].clone() for k in ("idx_cls", "idx_bbox")}
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = tg.run(x)
            torch.cuda.current_stream().wait_stream(side)
            assert got.sizes == ref.sizes
            for a, b in zip(got.tensors(), want):
                if a.dtype == torch.int64 and a.dim() == 2:       # ERS index lists: only the first count entries are defined
                    continue
                assert torch.equal(a, b), (rep, hw)
            cnt = got.ers["counts"].cpu()
            for n in range(2):
                for k, name in enumerate(("idx_cls", "idx_bbox")):
                    assert torch.equal(got.ers[name][n, :int(cnt[n, k])], want_idx[name][n, :int(cnt[n, k])]), (rep, hw, n, name)
            if rep == 1 and hw == (768, 1344):
                with torch.no_grad():
                    rc, rb = O.gfl_forward(tsd, x.cpu())
                rc, rb = O.flatten_levels(rc), O.flatten_levels(rb)
                for n in range(2):
                    ic, ib, _, _ = O.ers_select_single(rc[n], rb[n])
                    assert torch.equal(got.ers["idx_cls"][n, :int(cnt[n, 0])].cpu(), ic), n
                    assert torch.equal(got.ers["idx_bbox"][n, :int(cnt[n, 1])].cpu(), ib), n
    assert len(tg.graphs) == 3


def test_full_size_teacher_logits_and_ers_sets_vs_oracle():
    """BASELINE size, one image: the teacher's head outputs against the CPU oracle (1e-3; observed ~1e-5) and the ERS
    index sets bit-exact -- through the Winograd + direct kernels at the real level sizes (100x168 ... 7x11)."""
    import bench
    from oracle import erd_oracle as O
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd).eval()
    x, _ = bench.synthetic_gpu_batch(1, seed=40, device=torch.device("cuda", 0))
    with torch.no_grad():
        t = model.teacher_pass(x)
        ref_cls, ref_bbox = O.gfl_forward(tsd, x.cpu())
    rc, rb = O.flatten_levels(ref_cls), O.flatten_levels(ref_bbox)
    assert float((t.t_cls.cpu() - rc).abs().max()) < 1e-3 * float(rc.abs().max())
    assert float((t.t_bbox.cpu() - rb).abs().max()) < 1e-3 * float(rb.abs().max())
    ic, ib, thr_c, thr_b = O.ers_select_single(rc[0], rb[0])
    cnt = t.ers["counts"].cpu()
    assert torch.equal(t.ers["idx_cls"][0, :int(cnt[0, 0])].cpu(), ic)
    assert torch.equal(t.ers["idx_bbox"][0, :int(cnt[0, 1])].cpu(), ib)
    print("full-size teacher: max |dlogit| %.2e, ERS sets %d / %d anchors identical" %
          (float((t.t_cls.cpu() - rc).abs().max()), len(ic), len(ib)))


def test_full_size_step_losses_and_gradients_vs_oracle():
    """BASELINE size (800x1333 -> 800x1344, 22 400 anchors), one image: the WHOLE step -- teacher, ERS, NMS, student,
    all five loss groups and every parameter gradient -- against the CPU oracle on the same procedural weights and
    demo_mm_inputs-style sample.  Losses 1e-3 (north_star's tolerance; observed 1e-5), ERS index sets bit-exact,
    gradients within 1e-3 in relative L2 norm (median over the 175 tensors and over all elements together).

    Which kernels may be Winograd was measured over all 16 placements (tests/diag/diag_wino_matrix.py): the teacher and
    the input gradients for free, the student's recorded layers cost 2e-4 -> 7e-4, the student's FROZEN trunk (layer1:
    three 64->64 convolutions) 1.5e-3 -- a perturbation of the student's earliest activations is amplified through every
    trainable layer behind it.  Default: trunk on the direct kernels (kernels.WINO_FROZEN_TRUNK off); the second half of
    the test pins the rejected setting (< 5e-3) so that the trade-off cannot drift silently.

    The absolute numbers belong to THIS input (seed 7): single ReLU decisions flip under any fp32 re-ordering and one flip
    moves every gradient tensor by ~1e-3 -- on either side: for seed 8 the fp32 CPU reference itself sits 3.4e-3 from an
    fp64 evaluation of the step and this implementation 7e-5 (tests/diag/diag_fp64_truth.py 8)."""
    import numpy as np
    from oracle import erd_oracle as O
    from erd_amd import kernels as K
    from erd_amd import parse_losses
    from e2e_util import make_samples
    tsd, ssd = f7_state_dicts()
    imgs, boxes, labels = O.synthetic_batch(1, 800, 1333, 40, seed=7)
    x, metas = O.preprocess(imgs)
    sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v) for k, v in ssd.items()}
    ref_losses, aux = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80, return_aux=True)
    O.parse_losses(ref_losses).backward()
    names = [k for k, v in sd.items() if O.trainable(k) and v.dtype == torch.float32]
    ref = {k: sd[k].grad.double() for k in names}

    def gpu_step(wino_trunk: bool):
        keep, K.WINO_FROZEN_TRUNK = K.WINO_FROZEN_TRUNK, wino_trunk
        try:
            model = build_erd(tsd, ssd)
            losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
            total, _ = parse_losses(losses)
            total.backward()
            params = dict(model.named_parameters())
            grads = {k: params[k].grad.detach().cpu().double() for k in names}
            with torch.no_grad():
                t = model.teacher_pass(x.cuda())
            return {k: [float(v.detach()) for v in vs] for k, vs in losses.items()}, grads, t
        finally:
            K.WINO_FROZEN_TRUNK = keep

    def dist(ga, gb):
        errs, num, den = [], 0.0, 0.0
        for k in names:
            a, b = ga[k], gb[k]
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            if float(b.norm()) > 1e-12:
                errs.append(float((a - b).norm() / b.norm()))
        return float(np.median(errs)), max(errs), (num / den) ** 0.5

    for wino_trunk, tol in ((False, 1e-3), (True, 5e-3)):
        losses, grads, t = gpu_step(wino_trunk)
        for k, vs in ref_losses.items():
            assert np.allclose(losses[k], [float(v) for v in vs], rtol=1e-3, atol=1e-7), (k, losses[k], vs)
        cnt = t.ers["counts"].cpu()
        assert torch.equal(t.ers["idx_cls"][0, :int(cnt[0, 0])].cpu(), aux["ers_cls"][0])
        assert torch.equal(t.ers["idx_bbox"][0, :int(cnt[0, 1])].cpu(), aux["ers_bbox"][0])
        med, mx, glob = dist(grads, ref)
        print("full-size step, student's frozen trunk on the %s kernels: %d gradient tensors vs oracle: rel L2 median %.2e max %.2e global %.2e"
              % ("Winograd" if wino_trunk else "direct", len(names), med, mx, glob))
        assert med < tol and glob < tol, (wino_trunk, med, glob, mx)
