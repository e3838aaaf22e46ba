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
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O


@pytest.fixture(scope="module")
def nets():
    tsd, ssd = f7_state_dicts()
    return tsd, ssd, build_erd(tsd, ssd)


def test_ers_index_sets_over_32_full_size_images(nets):
    from erd_amd import kernels as K
    tsd, ssd, model = nets
    model.eval()
    nimg, bs = 32, 4
    stats = {True: [0, 0], False: [0, 0]}            # teacher on Winograd / direct: [images with a differing set, differing anchors]
    worst_margin = 1.0
    for b0 in range(0, nimg, bs):
        imgs, _, _ = O.synthetic_batch(bs, 800, 1333, 40, seed=100 + b0)
        x, _ = O.preprocess(imgs)
        with torch.no_grad():
            ref_cls, ref_bbox = O.gfl_forward(tsd, x)
        rc, rb = O.flatten_levels(ref_cls), O.flatten_levels(ref_bbox)
        for wino in (True, False):
            keep, K.WINO_TEACHER = K.WINO_TEACHER, wino
            try:
                with torch.no_grad():
                    t = model.teacher_pass(x.cuda())
            finally:
                K.WINO_TEACHER = keep
            cnt = t.ers["counts"].cpu()
            for i in range(bs):
                ic, ib, thr_c, thr_b = O.ers_select_single(rc[i], rb[i])
                gc = t.ers["idx_cls"][i, :int(cnt[i, 0])].cpu()
                gb = t.ers["idx_bbox"][i, :int(cnt[i, 1])].cpu()
                diff = set(ic.tolist()) ^ set(gc.tolist()) | set(ib.tolist()) ^ set(gb.tolist())
                if diff:
                    stats[wino][0] += 1
                    stats[wino][1] += len(diff)
                    # a tolerated difference is an anchor ON the threshold: its statistic within 1e-5 (relative) of it
                    mc = rc[i].sigmoid().max(-1)[0]
                    mb = rb[i].max(-1)[0]
                    for a in diff:
                        m = min(abs(float(mc[a]) - thr_c) / abs(thr_c), abs(float(mb[a]) - thr_b) / abs(thr_b))
                        worst_margin = min(worst_margin, m)
                        assert m < 1e-5, (wino, b0 + i, a, m)
    print("ERS sets vs the CPU oracle over %d full-size images: Winograd teacher %d images (%d anchors) differ, direct teacher %d (%d)"
          % (nimg, stats[True][0], stats[True][1], stats[False][0], stats[False][1]))
    assert stats[True][0] <= 1 and stats[False][0] <= 1, stats
    assert stats[True][0] <= stats[False][0] + 1, stats       # the default (Winograd) teacher is not worse than the direct one


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


def test_full_size_gradients_anchored_to_fp64(nets):
    """Four full-size steps (seeds 7-10), each evaluated three times: the oracle in fp64 (the truth), the oracle in fp32 (the
    reference's own arithmetic) and the HIP path.  Measured (profiles/r02 notes), relative L2 distance to fp64 as
    median over the 175 gradient tensors / all elements / worst tensor:
        seed   cpu fp32                      hip
          7    4.7e-4  4.1e-4  2.5e-3        6.8e-4  4.2e-4  2.3e-3
          8    3.4e-3  1.8e-3  4.8e-3        5.7e-4  5.9e-4  2.3e-3
          9    4.9e-4  5.1e-4  3.1e-3        1.1e-3  7.8e-4  3.2e-3
         10    8.7e-4  5.4e-4  1.7e-3        9.0e-4  5.9e-4  1.8e-3
    (The worst-tensor column read 3.5e-3 / 6.3e-3 / 1.1e-2 on seeds 8-10 until the Winograd cover stopped storing some
    tiles twice: the fused column sums of an input-gradient launch -- a BN bias gradient -- counted those pixels double.
    This test's bound D was loose enough to pass with the bug; test_winograd_cover_stores_every_pixel_exactly_once is the
    direct check, and D is now the measured 1.25x.)
    A ReLU whose pre-activation is ~1e-7 takes one side in one fp32 summation order and the other side in another; ONE such
    flip moves every gradient tensor upstream by ~1e-3 and a small-norm one (a BN bias gradient: a sum over a map with heavy
    cancellation) by up to 1e-2.  Which implementation owns the flip changes with the seed: the fp32 CPU reference on seed 8,
    this one on seed 9.  A per-seed bound "hip <= 1.25 x cpu" is therefore false for EITHER implementation on some seed; what
    holds, and is asserted:
      A. every seed: losses within 1e-3 of the fp32 reference and of fp64; the WHOLE gradient within 1e-3 of fp64;
      B. every seed: median per-tensor distance <= 1.5e-3, and <= max(1e-3, 1.25 x the reference's) on at least 3 of 4 seeds;
      C. over the seeds: this implementation's worst median / worst whole-gradient distance is not above the reference's worst;
      D. worst single tensor: <= 5e-3 on every seed and within 1.25x of the reference's worst tensor of the same seed."""
    from erd_amd import parse_losses
    tsd, ssd, _ = nets
    names = [k for k, v in ssd.items() if O.trainable(k) and v.dtype == torch.float32]

    def dist(ga, gb):
        errs, num, den = [], 0.0, 0.0
        for k in names:
            a, b = ga[k], gb[k]
            num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
            if float(b.norm()) > 1e-12:
                errs.append(float((a - b).norm() / b.norm()))
        return float(np.median(errs)), (num / den) ** 0.5, max(errs)

    cpu_rows, hip_rows = [], []
    for seed in (7, 8, 9, 10):
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
        g32, l32 = oracle(torch.float32)
        model = build_erd(tsd, ssd)
        losses = model(x.cuda(), make_samples(boxes, labels, metas), mode="loss")
        parse_losses(losses)[0].backward()
        p = dict(model.named_parameters())
        gh = {k: p[k].grad.detach().cpu().double() for k in names}
        for k, vs in l32.items():
            got = [float(v.detach()) for v in losses[k]]
            assert np.allclose(got, vs, rtol=1e-3, atol=1e-7) and np.allclose(got, l64[k], rtol=1e-3, atol=1e-7), (seed, k, got, vs)
        cpu, hip = dist(g32, g64), dist(gh, g64)
        cpu_rows.append(cpu); hip_rows.append(hip)
        print("seed %d, %d gradient tensors, rel L2 to fp64 (median / all elements / worst tensor): cpu fp32 %.2e %.2e %.2e | hip %.2e %.2e %.2e"
              % ((seed, len(names)) + cpu + hip))
        del model
    cpu, hip = np.array(cpu_rows), np.array(hip_rows)
    assert (hip[:, 1] <= 1e-3).all(), hip[:, 1]                                                     # A
    assert (hip[:, 0] <= 1.5e-3).all() and int((hip[:, 0] <= np.maximum(1e-3, 1.25 * cpu[:, 0])).sum()) >= 3, (hip[:, 0], cpu[:, 0])   # B
    assert hip[:, 0].max() <= cpu[:, 0].max() and hip[:, 1].max() <= cpu[:, 1].max(), (hip, cpu)   # C
    assert (hip[:, 2] <= 5e-3).all() and (hip[:, 2] <= 1.25 * cpu[:, 2]).all(), (hip[:, 2], cpu[:, 2])   # D
