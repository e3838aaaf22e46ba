"""CPU: COCO bbox mAP restatement (erd_amd/evaluation.py; pycocotools is absent -> unpinned) against known answers
that follow from COCOeval's published definition."""
import numpy as np
import pytest

from erd_amd.evaluation import CocoBBoxEval, split_map


def gt_of(anns, n_img=2, cats=(1, 2, 3)):
    return dict(images=[dict(id=i, width=640, height=480) for i in range(n_img)],
                categories=[dict(id=c, name=f"c{c}") for c in cats],
                annotations=[dict(id=k + 1, image_id=a[0], category_id=a[1], bbox=list(a[2]), area=a[2][2] * a[2][3],
                                  iscrowd=a[3] if len(a) > 3 else 0) for k, a in enumerate(anns)])


def xyxy(b):
    return [b[0], b[1], b[0] + b[2], b[1] + b[3]]


def test_perfect_detections_score_one():
    anns = [(0, 1, (10, 10, 100, 50)), (0, 2, (200, 100, 40, 40)), (1, 1, (5, 5, 300, 200)), (1, 3, (50, 60, 20, 25))]
    ev = CocoBBoxEval(gt_of(anns))
    for img in (0, 1):
        a = [x for x in anns if x[0] == img]
        ev.add_predictions(img, np.array([xyxy(x[2]) for x in a]), np.ones(len(a)), np.array([x[1] - 1 for x in a]))
    s = ev.evaluate()
    assert s["bbox_mAP"] == pytest.approx(1.0) and s["bbox_mAP_50"] == pytest.approx(1.0) and s["AR@100"] == pytest.approx(1.0)
    assert all(v == pytest.approx(1.0) for v in ev.classwise().values())
    assert s["bbox_mAP_s"] == pytest.approx(1.0) and s["bbox_mAP_l"] == pytest.approx(1.0)      # 20x25 small, 300x200 large


def test_higher_scored_false_positive_halves_precision():
    ev = CocoBBoxEval(gt_of([(0, 1, (10, 10, 100, 100))]), cat_ids=[1])
    ev.add_predictions(0, np.array([[300, 300, 400, 400], [10, 10, 110, 110]]), np.array([1.0, 0.9]), np.array([0, 0]))
    s = ev.evaluate()
    assert s["bbox_mAP"] == pytest.approx(0.5) and s["AR@100"] == pytest.approx(1.0) and s["AR@1"] == pytest.approx(0.0)


def test_iou_thresholds_count():
    # detection shifted so that IoU = 80*100 / (2*100*100 - 80*100) = 0.6667 -> matched at .50 .. .65 (4 of 10 thresholds)
    ev = CocoBBoxEval(gt_of([(0, 1, (0, 0, 100, 100))]), cat_ids=[1])
    ev.add_predictions(0, np.array([[20, 0, 120, 100]]), np.array([0.7]), np.array([0]))
    s = ev.evaluate()
    assert s["bbox_mAP"] == pytest.approx(0.4) and s["bbox_mAP_50"] == pytest.approx(1.0) and s["bbox_mAP_75"] == pytest.approx(0.0)


def test_crowd_region_absorbs_detections_without_penalty():
    anns = [(0, 1, (0, 0, 200, 200), 1), (0, 1, (300, 300, 50, 50), 0)]
    ev = CocoBBoxEval(gt_of(anns), cat_ids=[1])
    # two detections inside the crowd region (ignored, neither TP nor FP) + the real object
    ev.add_predictions(0, np.array([[10, 10, 60, 60], [100, 100, 150, 150], [300, 300, 350, 350]]),
                       np.array([0.99, 0.98, 0.5]), np.array([0, 0, 0]))
    s = ev.evaluate()
    assert s["bbox_mAP"] == pytest.approx(1.0)
    # without the crowd annotation the same detections are false positives in front of the true positive
    ev2 = CocoBBoxEval(gt_of(anns[1:]), cat_ids=[1])
    ev2.add_predictions(0, np.array([[10, 10, 60, 60], [100, 100, 150, 150], [300, 300, 350, 350]]),
                        np.array([0.99, 0.98, 0.5]), np.array([0, 0, 0]))
    assert ev2.evaluate()["bbox_mAP"] == pytest.approx(1 / 3)


def test_max_dets_area_ranges_and_old_new_split():
    anns = [(0, 1, (0, 0, 20, 20)), (0, 1, (100, 100, 20, 20)), (1, 2, (0, 0, 200, 200))]
    ev = CocoBBoxEval(gt_of(anns, cats=(1, 2)))
    ev.add_predictions(0, np.array([xyxy(anns[0][2]), xyxy(anns[1][2])]), np.array([0.9, 0.8]), np.array([0, 0]))
    ev.add_predictions(1, np.array([[0, 0, 100, 200]]), np.array([0.9]), np.array([1]))           # IoU 0.5 with the 200x200 box
    s = ev.evaluate()
    assert s["AR@1"] == pytest.approx((0.5 + 0.1) / 2)        # class 1: one of two objects; class 2: matched at 1 of 10 thresholds
    assert s["bbox_mAP_s"] == pytest.approx(1.0) and s["bbox_mAP_m"] == -1.0
    assert s["bbox_mAP_l"] == pytest.approx(0.1)
    sp = split_map(ev, old_cat_ids=[1])
    assert sp["old_mAP"] == pytest.approx(1.0) and sp["new_mAP"] == pytest.approx(0.1) and sp["all_mAP"] == pytest.approx(0.55)


def test_label_to_category_mapping_and_empty():
    ev = CocoBBoxEval(gt_of([(0, 3, (0, 0, 50, 50))]), cat_ids=[3, 1])       # label 0 -> category 3
    ev.add_predictions(0, np.array([[0, 0, 50, 50]]), np.array([0.3]), np.array([0]))
    ev.add_predictions(1, np.zeros((0, 4)), np.zeros(0), np.zeros(0, dtype=int))
    s = ev.evaluate()
    assert s["bbox_mAP"] == pytest.approx(1.0)
    assert np.isnan(ev.classwise()["c1"])
