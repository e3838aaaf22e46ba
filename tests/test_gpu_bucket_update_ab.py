"""GPU: the per-bucket update of ERDTrainer (engine.BucketedGradSync on_bucket -- SGD on a bucket's slice, its BN folds, its prepared
weight buffers, on a side stream, while the rest of the backward pass still runs) against the same trainer with the update in one
piece at the step boundary (ERD_BUCKET_UPDATE=0), from identical state, step by step: losses, the flat gradient and the parameters.

test_per_bucket_update_equals_the_one_launch_update (test_gpu_e2e.py) rebuilds SGD from the gradient the step LEFT -- a backward
launch that read already-updated weights would pass it.  Here such a race changes the gradient itself: an input-gradient
launch that read weights one update ahead is off by lr x gradient ~ 1e-3 relative, four orders above what is allowed.  The step's
float atomics (fused column sums) make two runs of the SAME configuration differ in the last bits, so the bound is taken from a
twin run of the reference configuration: |A - B| <= 4 |A - A'| + 1e-7 |A|.  TWO steps: from the third step on that 1e-8 noise flips a
discrete decision of the path now and then (an ERS threshold / ATSS tie: twin runs of one configuration then sit 1e-5 apart, 2e-3 a
step later -- tools/dbg/bucket_ab_probe.py), which says nothing about the update; a race shows in the first step's gradient.  Variants: 1 MB buckets (many early releases), the A/B
switches of functional.py that change which stream holds what (ERD_RES_LAYER=0, ERD_HEAD_TRAIL=1), and the whole run under a
NON-default current stream (the update stream must be ordered behind the streams of the moment, not those of construction).
Reference path: mmengine's OptimWrapper.update_params behind DDP's bucketed all-reduce (configs/_base_/schedules/schedule_1x.py,
default_runtime.py:14)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
STEPS = 2
LAG = 40_000_000          # spin cycles (tens of milliseconds)

from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O


def _batches():
    out = []
    for seed in (0, 1):
        imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=seed)
        x, metas = O.preprocess(imgs)
        out.append((x.cuda(), make_samples(boxes, labels, metas)))
    return out


def _run(bucket_update: bool, bucket_mb, steps, stream=None, lag=LAG):
    """`steps` optimisation steps from the fixture's state; per step: (loss, flat gradient, flat parameters).  `lag`: a spin kernel
    queued on the current stream in front of every step -- on this small model the host issues slower than the GPU executes, so an
    update that is NOT ordered behind its producers would still find them finished; with the GPU a step behind the host it runs
    first (tools/dbg/bucket_ab_negctl.py: an unordered update stream breaks the bound by orders of magnitude)"""
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    batches = _batches()
    old = os.environ.get("ERD_BUCKET_UPDATE")
    os.environ["ERD_BUCKET_UPDATE"] = "1" if bucket_update else "0"
    try:
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            model = build_erd(tsd, ssd)
            tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=3, bucket_mb=bucket_mb)
            assert tr.bucket_update == bucket_update
            rec = []
            for i in range(steps):
                if lag:
                    torch.cuda._sleep(int(lag))
                out = tr.train_step(*batches[i % 2], next_batch=batches[(i + 1) % 2])
                tr.flush()
                torch.cuda.synchronize()
                loss = float(out["loss"])
                rec.append((loss, tr.flat.grad.clone(), tr.flat.data.clone()))
                if bucket_update:
                    assert tr.sync.late_buckets == 0 and tr.sync.issued_in_backward >= len(tr.flat.buckets) - 1
        return rec
    finally:
        if old is None:
            os.environ.pop("ERD_BUCKET_UPDATE", None)
        else:
            os.environ["ERD_BUCKET_UPDATE"] = old


def _compare(ref, twin, got, what):
    for i, ((l0, g0, d0), (l1, g1, d1), (l2, g2, d2)) in enumerate(zip(ref, twin, got)):
        for name, a, a1, b in (("gradient", g0, g1, g2), ("parameters", d0, d1, d2)):
            noise = float((a - a1).double().norm())
            diff = float((a - b).double().norm())
            bound = 4.0 * noise + 1e-7 * float(a.double().norm())
            assert diff <= bound, f"{what}: step {i} {name}: |A - B| = {diff:.3e} > {bound:.3e} (twin-run noise {noise:.3e})"
        assert abs(l0 - l2) <= 4.0 * abs(l0 - l1) + 2e-6 * abs(l0), (what, i, l0, l1, l2)


@pytest.fixture(scope="module")
def reference_runs():
    """the update in one piece at the step boundary, twice (the noise floor of the float atomics)"""
    return _run(False, 1, STEPS), _run(False, 1, STEPS)


@pytest.mark.parametrize("bucket_mb", [1, 25])
def test_per_bucket_update_leaves_the_same_trajectory_as_the_update_at_the_step_boundary(reference_runs, bucket_mb):
    ref, twin = reference_runs
    _compare(ref, twin, _run(True, bucket_mb, STEPS), f"bucket_mb={bucket_mb}")


def test_per_bucket_update_under_a_caller_owned_stream(reference_runs):
    ref, twin = reference_runs
    s = torch.cuda.Stream()
    _compare(ref, twin, _run(True, 1, STEPS, stream=s), "non-default current stream")
    torch.cuda.current_stream().wait_stream(s)


@pytest.mark.parametrize("flag", ["RES_LAYER_NODE", "HEAD_TRAIL"])
def test_per_bucket_update_with_the_stream_layout_switches(reference_runs, flag):
    from erd_amd import functional as Fn
    ref, twin = reference_runs
    old = getattr(Fn, flag)
    setattr(Fn, flag, not old)           # ERD_RES_LAYER=0: per-block autograd nodes; ERD_HEAD_TRAIL=1: the towers' weight gradients trail too
    try:
        _compare(ref, twin, _run(True, 1, STEPS), f"{flag}={not old}")
    finally:
        setattr(Fn, flag, old)
