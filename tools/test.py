#!/usr/bin/env python3
"""Evaluate a detector checkpoint: COCO bbox mAP (+ the old / new class split of an incremental run).

    python tools/test.py CONFIG CHECKPOINT [--cfg-options k=v ...] [--old-classes 40] [--batch-size 4] [--out results.json]

Reference: tools/test.py + CocoMetric (mmdet/evaluation/metrics/coco_metric.py); the images go through the GPU
pipeline without flipping, detections are rescaled to the original image (`rescale=True`), the metric is
erd_amd.evaluation.CocoBBoxEval (COCOeval restated, unpinned)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Test (and eval) a detector (erd_amd)")
    ap.add_argument("config")
    ap.add_argument("checkpoint")
    ap.add_argument("--cfg-options", nargs="+")
    ap.add_argument("--old-classes", type=int, default=None, help="number of old categories for the old/new mAP split "
                    "(default: ori_setting.ori_num_classes of the config)")
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--max-images", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)

    import torch
    from train import parse_cfg_options
    from erd_amd import Config, MODELS
    from erd_amd.datasets import CocoAnnotations, GpuDetPipeline
    from erd_amd.evaluation import CocoBBoxEval, split_map
    from erd_amd.runner import load_checkpoint

    cfg = Config.fromfile(a.config)
    cfg.merge_from_dict(parse_cfg_options(a.cfg_options))
    if cfg.model.get("latest_model_flag") is not None:
        cfg.model.latest_model_flag = False            # the checkpoint carries the teacher copy (or none is needed to test)
    model = MODELS.build(cfg.model).cuda().eval()
    own = model.state_dict()
    sd = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
    sd = sd.get("state_dict", sd)
    sd = {k: v for k, v in sd.items() if k in own}                                   # a student-only test ignores ori_model.*
    missing = [k for k in own if k not in sd and not k.startswith("ori_model.")]
    if missing:
        raise RuntimeError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    model.load_state_dict({**{k: v for k, v in own.items() if k not in sd}, **sd}, strict=True)

    dcfg = cfg.test_dataloader.dataset
    root = dcfg.get("data_root", "")
    gt = json.load(open(os.path.join(root, dcfg["ann_file"])))
    ann = CocoAnnotations(gt, (dcfg.get("metainfo") or {}).get("classes"),
                          data_prefix=os.path.join(root, (dcfg.get("data_prefix") or {}).get("img", "")), test_mode=True)
    scale = next((t["scale"] for t in dcfg.get("pipeline", []) if t.get("type") == "Resize"), (1333, 800))
    pipe = GpuDetPipeline(ann, scale=tuple(scale), flip_prob=0.0)
    ev = CocoBBoxEval(gt, cat_ids=ann.cat_ids)
    n = len(ann) if a.max_images is None else min(len(ann), a.max_images)
    results = []
    from erd_amd.datasets import pinned, prefetch_map
    batches = [list(range(b0, min(n, b0 + a.batch_size))) for b0 in range(0, n, a.batch_size)]
    decode = lambda idx: (idx, [pinned(im) for im in pipe.decode(idx)])
    for idx, imgs in prefetch_map(decode, batches, int(cfg.test_dataloader.get("num_workers", 0)), 2):
        x, samples = pipe.assemble(idx, imgs)
        out = model(x, samples, mode="predict")
        for i, d in zip(idx, out):
            p = d.pred_instances
            img_id = ann.get_data_info(i)["img_id"]
            bb, sc, lb = p.bboxes.cpu().numpy(), p.scores.cpu().numpy(), p.labels.cpu().numpy()
            ev.add_predictions(img_id, bb, sc, lb)
            for (x1, y1, x2, y2), s, l in zip(bb.tolist(), sc.tolist(), lb.tolist()):
                results.append(dict(image_id=img_id, category_id=ann.cat_ids[l], bbox=[x1, y1, x2 - x1, y2 - y1], score=s))
    stats = ev.evaluate()
    n_old = a.old_classes if a.old_classes is not None else (cfg.model.get("ori_setting") or {}).get("ori_num_classes")
    if n_old:
        stats.update(split_map(ev, ann.cat_ids[:n_old]))
    for k, v in stats.items():
        print(f"{k:12s} {v:.4f}")
    if a.out:
        json.dump(dict(stats=stats, classwise=ev.classwise(), results=results), open(a.out, "w"))
    return stats


if __name__ == "__main__":
    main()
