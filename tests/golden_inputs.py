"""Seeded input generators shared by ``oracle/gen_golden.py`` (which feeds them to the
REAL reference source in the build container) and by the tests (which feed the same
inputs to the oracle restatement and to the HIP path).  numpy PCG64 -> float32, so
the inputs are bit-identical on every machine; only *expected outputs* are stored
in ``tests/golden/*.npz``."""
from __future__ import annotations

import numpy as np
import torch


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def randn(seed, *shape, scale=1.0, shift=0.0):
    return torch.from_numpy((shift + scale * _rng(seed).standard_normal(shape)).astype(np.float32))


def rand(seed, *shape):
    return torch.from_numpy(_rng(seed).random(shape).astype(np.float32))


def randint(seed, lo, hi, *shape):
    return torch.from_numpy(_rng(seed).integers(lo, hi, size=shape).astype(np.int64))


def rand_boxes(seed, n, w=1333.0, h=800.0, min_size=4.0):
    """n xyxy boxes inside (w,h) with sides >= min_size."""
    r = _rng(seed)
    cx, cy = r.random(n) * w, r.random(n) * h
    bw = min_size + r.random(n) * 0.5 * w
    bh = min_size + r.random(n) * 0.5 * h
    b = np.stack([np.clip(cx - bw / 2, 0, w), np.clip(cy - bh / 2, 0, h),
                  np.clip(cx + bw / 2, 0, w), np.clip(cy + bh / 2, 0, h)], 1)
    return torch.from_numpy(b.astype(np.float32))


# ---- F1: leaf losses ---------------------------------------------------------------------
def f1_inputs():
    n, C = 300, 40
    d = dict(
        qfl_pred=randn(101, n, C, scale=2.0, shift=-2.0),
        qfl_label=randint(102, 0, C + 8, n).clamp(max=C),     # ~17 % background (= C)
        qfl_score=rand(103, n),
        qfl_weight=(rand(104, n) > 0.1).float(),
        dfl_pred=randn(105, 4 * 96, 17, scale=1.5),
        dfl_label=rand(106, 4 * 96) * 15.9,
        dfl_weight=rand(107, 4 * 96),
        kd_pred=randn(108, 4 * 64, 17, scale=2.0),
        kd_soft=randn(109, 4 * 64, 17, scale=2.0),
        kd_weight=rand(110, 4 * 64),
        box_a=rand_boxes(111, 128, 100.0, 60.0),
        box_b=rand_boxes(112, 128, 100.0, 60.0),
        box_w=rand(113, 128),
        pts=rand(114, 128, 2) * 50.0,
        dist=rand(115, 128, 4) * 20.0,
        l2_a=randn(116, 77, 40),
        l2_b=randn(117, 77, 40),
    )
    return d


# ---- F3: ATSS ------------------------------------------------------------------------------
ATSS_CASES = [dict(seed=300 + i, num_gt=g) for i, g in enumerate((1, 3, 7, 20))]


def atss_case(case, w=1333.0, h=800.0, num_classes=40):
    return rand_boxes(case["seed"], case["num_gt"], w, h, min_size=8.0), \
        randint(case["seed"] + 50, 0, num_classes, case["num_gt"])


# ---- F4: ERS ---------------------------------------------------------------------------------
def ers_inputs(seed, A=22400, C=40):
    """teacher-like logits: cls ~ N(-4,1.5^2) with a sparse set of confident anchors,
    bbox raw logits ~ N(0,1) with peaked rows."""
    r = _rng(seed)
    cls = (-4.0 + 1.5 * r.standard_normal((A, C))).astype(np.float32)
    hot = r.integers(0, A, size=A // 40)
    cls[hot, r.integers(0, C, size=hot.shape[0])] += (4.0 + 3.0 * r.random(hot.shape[0])).astype(np.float32)
    bbox = r.standard_normal((A, 68)).astype(np.float32)
    hot = r.integers(0, A, size=A // 30)
    bbox[hot, r.integers(0, 68, size=hot.shape[0])] += (2.0 + 3.0 * r.random(hot.shape[0])).astype(np.float32)
    return torch.from_numpy(cls), torch.from_numpy(bbox)


# ---- F5: NMS ------------------------------------------------------------------------------------
def nms_inputs(seed, n=600, ncls=40):
    b = rand_boxes(seed, n, 1333.0, 800.0, min_size=16.0)
    s = rand(seed + 1, n)
    ids = randint(seed + 2, 0, ncls, n)
    return b, s, ids


# ---- F6: head-level --------------------------------------------------------------------------------
def f6_inputs(N=2, H=256, W=256, c_old=40, c_all=80):
    sizes = [(H // s, W // s) for s in (8, 16, 32, 64, 128)]
    t_cls = [randn(600 + l, N, c_old, h, w, scale=1.5, shift=-3.0) for l, (h, w) in enumerate(sizes)]
    t_bbox = [randn(610 + l, N, 68, h, w, scale=1.5) for l, (h, w) in enumerate(sizes)]
    s_cls = [randn(620 + l, N, c_all, h, w, scale=1.5, shift=-3.0) for l, (h, w) in enumerate(sizes)]
    s_bbox = [randn(630 + l, N, 68, h, w, scale=1.5) for l, (h, w) in enumerate(sizes)]
    # make a few teacher anchors confident so ERS/NMS have something to chew on
    for l in range(3):
        t_cls[l][:, 3, ::5, ::7] += 5.0
        t_bbox[l][:, 11, ::6, ::5] += 4.0
    gtb = [rand_boxes(640 + i, 3 + 2 * i, float(W - 3), float(H - 5), min_size=12.0) for i in range(N)]
    gtl = [randint(650 + i, 0, c_all - c_old, 3 + 2 * i) for i in range(N)]
    metas = [dict(img_shape=(H - 5, W - 3), pad_shape=(H, W), batch_input_shape=(H, W)) for _ in range(N)]
    return sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas


def f10_inputs(case):
    """F6's pyramids with empty ground truth: case 0 = image 0 has no boxes, case 1 = no image has any (no positive
    anchor in the whole batch: every supervised box / DFL term takes the reference's `pred.sum() * 0` branch)."""
    sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas = f6_inputs()
    empty_b, empty_l = torch.zeros((0, 4)), torch.zeros((0,), dtype=torch.long)
    gtb = [empty_b, gtb[1]] if case == 0 else [empty_b, empty_b.clone()]
    gtl = [empty_l, gtl[1]] if case == 0 else [empty_l, empty_l.clone()]
    return sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas


# ---- F8: inference post-processing ---------------------------------------------------------------------
def f8_inputs(case):
    """case 0: 2 images, 256x256 pyramid, 80 classes, ~half of the scores above 0.05 (top-k cut on levels 0-2),
    img_shape smaller than the pad (boxes clamp to zero size and are filtered), rescale by (1.6, 1.5).
    case 1: 1 image, 128x96, 40 classes, few candidates (no level reaches nms_pre), no rescale."""
    if case == 0:
        N, H, W, C = 2, 256, 256, 80
        sizes = [(H // s, W // s) for s in (8, 16, 32, 64, 128)]
        cls = [randn(800 + l, N, C, h, w, scale=1.5, shift=-3.0) for l, (h, w) in enumerate(sizes)]
        bbox = [randn(810 + l, N, 68, h, w, scale=1.5) for l, (h, w) in enumerate(sizes)]
        metas = [dict(img_shape=(200, 220), pad_shape=(H, W), batch_input_shape=(H, W), scale_factor=(1.6, 1.5))
                 for _ in range(N)]
        return cls, bbox, metas, True
    N, H, W, C = 1, 128, 96, 40
    sizes = [(-(-H // s), -(-W // s)) for s in (8, 16, 32, 64, 128)]
    cls = [randn(820 + l, N, C, h, w, scale=1.2, shift=-6.0) for l, (h, w) in enumerate(sizes)]
    bbox = [randn(830 + l, N, 68, h, w, scale=2.5) for l, (h, w) in enumerate(sizes)]
    metas = [dict(img_shape=(H, W), pad_shape=(H, W), batch_input_shape=(H, W), scale_factor=(1.0, 1.0))]
    return cls, bbox, metas, False
