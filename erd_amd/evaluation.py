"""COCO bounding-box mAP (second half): the numbers the ERD paper reports for the old / new class
splits -- AP@[.50:.95], AP50, AP75, AP small/medium/large, AR@1/10/100, and the class-wise table.

The reference delegates this to pycocotools' ``COCOeval`` through ``mmdet/evaluation/metrics/coco_metric.py``;
pycocotools is not in this image, so the procedure is restated here from its published definition (cocoeval.py):
per (category, image): detections sorted by score (stable), at most maxDets; each is matched to the still-unmatched
ground truth of highest IoU >= threshold (crowd regions may be matched repeatedly and use intersection / detection
area; a match to an ignored ground truth makes the detection ignored); ground truth outside the area range is
ignored, unmatched detections outside the range are ignored too; precision is made monotone and sampled at 101 recall
points.  **Unpinned** against pycocotools (no copy to run here): `tests/test_evaluation_cpu.py` holds known answers
instead.  Host-side numpy, as in the reference.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}
MAX_DETS = (1, 10, 100)


def _iou_xywh(d: np.ndarray, g: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    """[D,4] x [G,4] boxes as (x, y, w, h); for crowd ground truth the union is the detection's own area."""
    if d.shape[0] == 0 or g.shape[0] == 0:
        return np.zeros((d.shape[0], g.shape[0]))
    dx2, dy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3]
    gx2, gy2 = g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    iw = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = iw * ih
    da, ga = d[:, 2] * d[:, 3], g[:, 2] * g[:, 3]
    union = np.where(crowd[None, :], da[:, None], da[:, None] + ga[None, :] - inter)
    return inter / np.maximum(union, 1e-12)


class CocoBBoxEval:
    """``gt``: COCO annotation dict (images / annotations / categories); ``cat_ids``: the categories to evaluate (the
    dataset's class order: label k of a prediction means cat_ids[k])."""

    def __init__(self, gt: dict, cat_ids: Optional[Sequence[int]] = None):
        self.cat_ids = list(cat_ids) if cat_ids is not None else sorted(c["id"] for c in gt["categories"])
        self.cat_names = {c["id"]: c["name"] for c in gt["categories"]}
        self.img_ids = [im["id"] for im in gt["images"]]
        self._gt = defaultdict(list)
        for a in gt["annotations"]:
            if a["category_id"] in set(self.cat_ids):
                self._gt[(a["image_id"], a["category_id"])].append(a)
        self._dt = defaultdict(list)

    def add_predictions(self, image_id: int, bboxes_xyxy: np.ndarray, scores: np.ndarray, labels: np.ndarray) -> None:
        """one image's detections in the reference's result format (xyxy -> xywh as CocoMetric.xyxy2xywh does)"""
        b = np.asarray(bboxes_xyxy, dtype=np.float64).reshape(-1, 4)
        for (x1, y1, x2, y2), s, l in zip(b, np.asarray(scores, dtype=np.float64), np.asarray(labels).astype(int)):
            self._dt[(image_id, self.cat_ids[l])].append((float(s), [x1, y1, x2 - x1, y2 - y1]))

    # -- per (image, category, area range) matching -----------------------------------------------------------
    def _evaluate_img(self, img_id, cat_id, rng, max_det):
        gts = self._gt.get((img_id, cat_id), [])
        dts = self._dt.get((img_id, cat_id), [])
        if not gts and not dts:
            return None
        g_ignore = np.array([bool(g.get("ignore", 0)) or bool(g.get("iscrowd", 0)) or
                             not (rng[0] <= g["area"] <= rng[1]) for g in gts], dtype=bool)
        gorder = np.argsort(g_ignore, kind="mergesort")                   # non-ignored first
        gts = [gts[i] for i in gorder]
        g_ignore = g_ignore[gorder]
        crowd = np.array([bool(g.get("iscrowd", 0)) for g in gts], dtype=bool)
        dorder = np.argsort([-s for s, _ in dts], kind="mergesort")[:max_det]
        dts = [dts[i] for i in dorder]
        db = np.array([b for _, b in dts], dtype=np.float64).reshape(-1, 4)
        gb = np.array([g["bbox"] for g in gts], dtype=np.float64).reshape(-1, 4)
        ious = _iou_xywh(db, gb, crowd)
        T, D, G = len(IOU_THRS), len(dts), len(gts)
        gtm = -np.ones((T, G), dtype=int)
        dtm = -np.ones((T, D), dtype=int)
        dt_ig = np.zeros((T, D), dtype=bool)
        for ti, t in enumerate(IOU_THRS):
            for di in range(D):
                best, m = min(t, 1 - 1e-10), -1
                for gi in range(G):
                    if gtm[ti, gi] >= 0 and not crowd[gi]:
                        continue
                    if m > -1 and not g_ignore[m] and g_ignore[gi]:
                        break                                  # ignored ground truth comes last: stop at the first one
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ignore[m]
                dtm[ti, di] = m
                gtm[ti, m] = di
        d_area = db[:, 2] * db[:, 3] if D else np.zeros(0)
        out_of_rng = ~((rng[0] <= d_area) & (d_area <= rng[1]))
        dt_ig = dt_ig | ((dtm < 0) & out_of_rng[None, :])
        return dict(scores=np.array([s for s, _ in dts]), dtm=dtm, dt_ig=dt_ig, n_gt=int((~g_ignore).sum()))

    # -- accumulate ----------------------------------------------------------------------------------------------
    def evaluate(self) -> Dict[str, float]:
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cat_ids):
            for a, rng in enumerate(AREA_RNG.values()):
                for m, max_det in enumerate(MAX_DETS):
                    evs = [e for e in (self._evaluate_img(i, cat, rng, max_det) for i in self.img_ids) if e is not None]
                    if not evs:
                        continue
                    n_gt = sum(e["n_gt"] for e in evs)
                    if n_gt == 0:
                        continue
                    scores = np.concatenate([e["scores"] for e in evs])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtm"] for e in evs], axis=1)[:, order]
                    dt_ig = np.concatenate([e["dt_ig"] for e in evs], axis=1)[:, order]
                    tps = np.cumsum((dtm >= 0) & ~dt_ig, axis=1, dtype=np.float64)
                    fps = np.cumsum((dtm < 0) & ~dt_ig, axis=1, dtype=np.float64)
                    for t in range(T):
                        tp, fp = tps[t], fps[t]
                        nd = len(tp)
                        rc = tp / n_gt
                        pr = tp / np.maximum(tp + fp, np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0.0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):           # monotone envelope
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds = np.searchsorted(rc, REC_THRS, side="left")
                        q = np.zeros(R)
                        for ri, pi in enumerate(inds):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.precision, self.recall = precision, recall

        def ap(iou=None, area="all", max_det=100):
            a, m = list(AREA_RNG).index(area), MAX_DETS.index(max_det)
            p = precision[:, :, :, a, m] if iou is None else precision[np.isclose(IOU_THRS, iou)][:, :, :, a, m]
            p = p[p > -1]
            return float(p.mean()) if p.size else -1.0

        def ar(area="all", max_det=100):
            a, m = list(AREA_RNG).index(area), MAX_DETS.index(max_det)
            r = recall[:, :, a, m]
            r = r[r > -1]
            return float(r.mean()) if r.size else -1.0

        stats = {"bbox_mAP": ap(), "bbox_mAP_50": ap(0.5), "bbox_mAP_75": ap(0.75), "bbox_mAP_s": ap(area="small"),
                 "bbox_mAP_m": ap(area="medium"), "bbox_mAP_l": ap(area="large"), "AR@1": ar(max_det=1),
                 "AR@10": ar(max_det=10), "AR@100": ar(), "AR_s@100": ar("small"), "AR_m@100": ar("medium"),
                 "AR_l@100": ar("large")}
        self.stats = stats
        return stats

    def classwise(self) -> Dict[str, float]:
        """AP@[.50:.95] per category (CocoMetric(classwise=True)); call evaluate() first"""
        out = {}
        for k, cat in enumerate(self.cat_ids):
            p = self.precision[:, :, k, 0, MAX_DETS.index(100)]
            p = p[p > -1]
            out[self.cat_names.get(cat, str(cat))] = float(p.mean()) if p.size else float("nan")
        return out


def split_map(ev: CocoBBoxEval, old_cat_ids: Iterable[int]) -> Dict[str, float]:
    """mean class-wise AP over the old and the new categories -- the two numbers an incremental-detection table shows"""
    cw = ev.classwise()
    old = {ev.cat_names[c] for c in old_cat_ids}
    o = [v for k, v in cw.items() if k in old and v == v]
    nw = [v for k, v in cw.items() if k not in old and v == v]
    return dict(old_mAP=float(np.mean(o)) if o else float("nan"), new_mAP=float(np.mean(nw)) if nw else float("nan"),
                all_mAP=float(np.mean(o + nw)) if (o or nw) else float("nan"))
