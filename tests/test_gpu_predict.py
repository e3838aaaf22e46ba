"""GPU: the inference post-processing through the C ABI (erd_predict_topk / erd_predict_nms)
against (a) the F8 fixture produced by the reference's GFLHead.predict_by_feat and (b) the oracle restatement
on seeded inputs: threshold, per-level top-k, ties, empty levels, rescale, size filter, NMS, max_per_img.

Labels and the detection set must match exactly; scores within 1e-6, boxes within 1e-3 px (GPU expf vs the host's).
Two detections whose scores differ by less than 2e-7 may swap places (the reference's sort is unstable anyway)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_inputs as G
from oracle import erd_oracle as O

TEST_CFG = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type="nms", iou_threshold=0.6),
                max_per_img=100)


@pytest.fixture(scope="module")
def head80():
    from erd_amd import MODELS
    assert torch.cuda.is_available()
    return lambda C=80, cfg=TEST_CFG: MODELS.build(dict(type="GFLHead", num_classes=C, in_channels=256,
                                                        test_cfg=dict(cfg))).cuda()


def assert_same_detections(got, want, box_tol=1e-3, score_tol=1e-6):
    gb, gs, gl = [t.detach().cpu() for t in got]
    wb, ws, wl = [torch.as_tensor(t) for t in want]
    assert gb.shape[0] == wb.shape[0], (gb.shape, wb.shape)
    assert (gs[:-1] >= gs[1:]).all()
    used = torch.zeros(len(wb), dtype=torch.bool)
    for i in range(len(gb)):
        # candidates: same label, score within tol; normally position i itself
        cand = ((wl == gl[i]) & ((ws - gs[i]).abs() <= score_tol) & ~used).nonzero().flatten()
        ok = [j for j in cand.tolist() if (wb[j] - gb[i]).abs().max() <= box_tol]
        assert ok, f"detection {i}: label {int(gl[i])} score {float(gs[i])} box {gb[i].tolist()} has no partner"
        j = min(ok, key=lambda j: abs(j - i))
        assert abs(j - i) == 0 or abs(float(ws[j] - ws[i])) <= 2e-7, (i, j)
        used[j] = True


@pytest.mark.parametrize("case", [0, 1])
def test_predict_by_feat_vs_reference_fixture(head80, golden, case):
    g = golden("f8_predict_unpinned_nms.npz")
    cls, bbox, metas, rescale = G.f8_inputs(case)
    head = head80(cls[0].shape[1])
    res = head.predict_by_feat([c.cuda() for c in cls], [b.cuda() for b in bbox], batch_img_metas=metas,
                               rescale=rescale)
    for i, r in enumerate(res):
        assert r.labels.dtype == torch.int64
        assert_same_detections((r.bboxes, r.scores, r.labels),
                               (g[f"c{case}_i{i}_bboxes"], g[f"c{case}_i{i}_scores"], g[f"c{case}_i{i}_labels"]))


@pytest.mark.parametrize("cfg", [
    dict(nms_pre=64, min_bbox_size=0, score_thr=0.3, nms=dict(type="nms", iou_threshold=0.5), max_per_img=20),
    dict(nms_pre=1000, min_bbox_size=-1, score_thr=0.05, nms=dict(type="nms", iou_threshold=0.6), max_per_img=300),
    dict(nms_pre=7, min_bbox_size=4, score_thr=0.0, nms=dict(type="nms", iou_threshold=0.9), max_per_img=100),
])
def test_predict_config_sweep_vs_oracle(head80, cfg):
    cls, bbox, metas, _ = G.f8_inputs(0)
    head = head80(80, cfg)
    res = head.predict_by_feat([c.cuda() for c in cls], [b.cuda() for b in bbox], batch_img_metas=metas, rescale=True)
    want = O.predict_by_feat(cls, bbox, metas, rescale=True, score_thr=cfg["score_thr"], nms_pre=cfg["nms_pre"],
                             min_bbox_size=cfg["min_bbox_size"], iou_thr=cfg["nms"]["iou_threshold"],
                             max_per_img=cfg["max_per_img"])
    for r, w in zip(res, want):
        assert_same_detections((r.bboxes, r.scores, r.labels), w)


def test_predict_ties_cut_by_topk_take_lowest_indices(head80):
    """every score equal (0.5): the top-k cut falls inside one giant tie; the lowest (anchor, class) indices win,
    exactly like a stable sort (the tie-select kernel)."""
    N, C, H, W = 2, 80, 128, 128
    sizes = [(H // s, W // s) for s in (8, 16, 32, 64, 128)]
    cls = [torch.zeros(N, C, h, w) for h, w in sizes]
    cls[1][1, :, 0, 0] = 2.0        # a few strictly larger ones on one level of one image
    bbox = [G.randn(900 + l, N, 68, h, w, scale=2.0) for l, (h, w) in enumerate(sizes)]
    metas = [dict(img_shape=(H, W), scale_factor=(1.0, 1.0)) for _ in range(N)]
    cfg = dict(nms_pre=100, min_bbox_size=0, score_thr=0.05, nms=dict(type="nms", iou_threshold=0.6), max_per_img=100)
    res = head80(80, cfg).predict_by_feat([c.cuda() for c in cls], [b.cuda() for b in bbox], batch_img_metas=metas)
    want = O.predict_by_feat(cls, bbox, metas, rescale=False, score_thr=0.05, nms_pre=100, min_bbox_size=0,
                             iou_thr=0.6, max_per_img=100)
    for r, w in zip(res, want):
        assert torch.equal(r.labels.cpu(), w[2])
        assert torch.allclose(r.scores.cpu(), w[1], atol=1e-6) and torch.allclose(r.bboxes.cpu(), w[0], atol=1e-3)


def test_predict_nothing_above_threshold(head80):
    N, C = 2, 40
    sizes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    cls = [torch.full((N, C, h, w), -20.0) for h, w in sizes]
    cls[0][1, 3, 2, 2] = 3.0        # image 1 gets exactly one detection, image 0 none
    bbox = [G.randn(910 + l, N, 68, h, w) for l, (h, w) in enumerate(sizes)]
    metas = [dict(img_shape=(64, 64), scale_factor=(2.0, 2.0)) for _ in range(N)]
    res = head80(40).predict_by_feat([c.cuda() for c in cls], [b.cuda() for b in bbox], batch_img_metas=metas,
                                     rescale=True)
    assert len(res[0].bboxes) == 0 and res[0].bboxes.shape == (0, 4) and len(res[0].labels) == 0
    want = O.predict_by_feat(cls, bbox, metas, rescale=True)
    assert len(res[1].bboxes) == 1
    assert_same_detections((res[1].bboxes, res[1].scores, res[1].labels), want[1])


def test_predict_full_size_vs_oracle_and_properties(head80):
    """BASELINE size: 800x1344, 80 classes, 2 images (22 400 anchors x 80 scores each)."""
    N, C, H, W = 2, 80, 800, 1344
    sizes, h, w = [], H // 8, W // 8
    for _ in range(5):
        sizes.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    cls = [G.randn(920 + l, N, C, a, b, scale=1.3, shift=-4.5) for l, (a, b) in enumerate(sizes)]
    bbox = [G.randn(930 + l, N, 68, a, b, scale=2.0) for l, (a, b) in enumerate(sizes)]
    metas = [dict(img_shape=(800, 1333), scale_factor=(1.333, 1.333)) for _ in range(N)]
    res = head80(80).predict_by_feat([c.cuda() for c in cls], [b.cuda() for b in bbox], batch_img_metas=metas,
                                     rescale=True)
    want = O.predict_by_feat(cls, bbox, metas, rescale=True)
    for r, wnt in zip(res, want):
        assert_same_detections((r.bboxes, r.scores, r.labels), wnt)
        b, s, l = r.bboxes.cpu(), r.scores.cpu(), r.labels.cpu()
        assert len(b) <= 100 and (s > 0.05).all() and (s[:-1] >= s[1:]).all()
        iou = O.bbox_overlaps(b, b)
        same = l[:, None] == l[None, :]
        iou = iou.masked_fill(~same, 0).fill_diagonal_(0)
        assert float(iou.max()) <= 0.6 + 1e-6       # no two kept boxes of a class overlap more than the threshold


def test_detector_predict_mode_matches_oracle_on_its_own_head_outputs():
    """GFLIncrementERD(mode='predict') == post-processing of its own mode='tensor' outputs (the 80-class student)."""
    import e2e_util as U
    tsd, ssd = U.f7_state_dicts()
    model = U.build_erd(tsd, ssd).eval()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=3)
    x, metas = O.preprocess(imgs)
    for m in metas:
        m["scale_factor"] = (0.8, 0.8)
    samples = U.make_samples(boxes, labels, metas)
    with torch.no_grad():
        cls, bbox = model(x.cuda(), samples, mode="tensor")
    out = model(x.cuda(), samples, mode="predict")
    want = O.predict_by_feat([c.cpu() for c in cls], [b.cpu() for b in bbox], metas, rescale=True)
    for d, w in zip(out, want):
        p = d.pred_instances
        assert_same_detections((p.bboxes, p.scores, p.labels), w)


def test_predict_results_feed_the_coco_evaluator():
    """mode='predict' -> CocoBBoxEval: detections of the model on its own synthetic ground truth boxes flow through the
    evaluator (old/new split table included); with the teacher's boxes fed back as predictions the metric is 1."""
    import e2e_util as U
    from erd_amd.evaluation import CocoBBoxEval, split_map
    tsd, ssd = U.f7_state_dicts()
    model = U.build_erd(tsd, ssd).eval()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=4)
    x, metas = O.preprocess(imgs)
    for m in metas:
        m["scale_factor"] = (1.0, 1.0)
    cats = [dict(id=10 + k, name=f"class{k}") for k in range(80)]
    anns, aid = [], 1
    for i, (b, l) in enumerate(zip(boxes, labels)):
        for bb, ll in zip(b.tolist(), l.tolist()):
            anns.append(dict(id=aid, image_id=i, category_id=10 + 40 + ll, bbox=[bb[0], bb[1], bb[2] - bb[0], bb[3] - bb[1]],
                             area=(bb[2] - bb[0]) * (bb[3] - bb[1]), iscrowd=0))
            aid += 1
    gt = dict(images=[dict(id=i, width=153, height=123) for i in range(2)], categories=cats, annotations=anns)
    out = model(x.cuda(), U.make_samples(boxes, labels, metas), mode="predict")
    ev = CocoBBoxEval(gt, cat_ids=[c["id"] for c in cats])
    for i, d in enumerate(out):
        p = d.pred_instances
        ev.add_predictions(i, p.bboxes.cpu().numpy(), p.scores.cpu().numpy(), p.labels.cpu().numpy())
    s = ev.evaluate()
    assert -1.0 <= s["bbox_mAP"] <= 1.0 and set(split_map(ev, [c["id"] for c in cats[:40]])) == {"old_mAP", "new_mAP", "all_mAP"}
    ev2 = CocoBBoxEval(gt, cat_ids=[c["id"] for c in cats])
    for i, (b, l) in enumerate(zip(boxes, labels)):
        ev2.add_predictions(i, b.numpy(), np.ones(len(b)), (l + 40).numpy())
    assert ev2.evaluate()["bbox_mAP"] == pytest.approx(1.0)
    assert split_map(ev2, [c["id"] for c in cats[:40]])["new_mAP"] == pytest.approx(1.0)
