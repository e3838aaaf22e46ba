"""GPU: the image side of the data pipeline -- PNG files on disk -> GpuDetPipeline.batch ->
normalised padded batch + data samples, against the test oracle (oracle/image_ops.py: OpenCV's 8-bit bilinear resize
restated, UNPINNED vs cv2) and the DetDataPreprocessor arithmetic.  Bit-exact pixels; boxes exactly as the
Resize / RandomFlip box rules give them."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import erd_oracle as O
from oracle import image_ops as I


def _make_dataset(tmp_path, sizes):
    from PIL import Image
    rng = np.random.RandomState(7)
    images, anns = [], []
    for i, (h, w) in enumerate(sizes):
        arr = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(arr).save(tmp_path / f"{i:04d}.png")                    # RGB on disk
        images.append(dict(id=100 + i, file_name=f"{i:04d}.png", width=w, height=h))
        for k in range(2):
            x, y = rng.uniform(0, w - 20), rng.uniform(0, h - 20)
            bw, bh = rng.uniform(8, w - x), rng.uniform(8, h - y)
            anns.append(dict(id=len(anns) + 1, image_id=100 + i, category_id=1 + k, bbox=[float(x), float(y), float(bw), float(bh)],
                             area=float(bw * bh), iscrowd=0))
    ds = dict(images=images, annotations=anns, categories=[dict(id=1, name="a"), dict(id=2, name="b")])
    json.dump(ds, open(tmp_path / "ann.json", "w"))
    return ds


def test_pipeline_batch_bit_exact_vs_oracle(tmp_path):
    from erd_amd.datasets import CocoAnnotations, GpuDetPipeline, load_image_bgr
    sizes = [(48, 64), (60, 45), (33, 80)]
    ds = _make_dataset(tmp_path, sizes)
    ann = CocoAnnotations(str(tmp_path / "ann.json"), classes=("a", "b"), data_prefix=str(tmp_path), min_size=0)
    pipe = GpuDetPipeline(ann, scale=(133, 80), flip_prob=0.5, seed=2)
    x, samples = pipe.batch([0, 1, 2])
    mean = torch.tensor(O.PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(O.PIXEL_STD).view(3, 1, 1)
    flips = []
    for k, s in enumerate(samples):
        bgr = load_image_bgr(str(tmp_path / f"{k:04d}.png"))
        m = s.metainfo
        flips.append(m["flip"])
        want_u8, sf = I.resize_flip(bgr, (133, 80), flip=m["flip"])
        nh, nw = want_u8.shape[:2]
        assert m["img_shape"] == (nh, nw) and m["scale_factor"] == pytest.approx(sf)
        ref = (torch.from_numpy(want_u8[:, :, ::-1].copy()).permute(2, 0, 1).float() - mean) / std      # BGR -> RGB, normalise
        got = x[k].cpu()
        assert torch.equal(got[:, :nh, :nw], ref), k
        assert float(got[:, nh:, :].abs().max() if nh < got.shape[1] else 0) == 0 and \
            float(got[:, :, nw:].abs().max() if nw < got.shape[2] else 0) == 0
        # boxes: scaled, clipped to the resized image, then mirrored
        d = ann.get_data_info(k)
        b = torch.tensor([i["bbox"] for i in d["instances"]], dtype=torch.float32) * torch.tensor([sf[0], sf[1], sf[0], sf[1]],
                                                                                                    dtype=torch.float32)
        b[:, 0::2].clamp_(0, nw); b[:, 1::2].clamp_(0, nh)
        if m["flip"]:
            b = torch.stack([nw - b[:, 2], b[:, 1], nw - b[:, 0], b[:, 3]], 1)
        assert torch.allclose(s.gt_instances.bboxes, b) and s.gt_instances.labels.tolist() == [0, 1]
        assert m["pad_shape"] == tuple(x.shape[2:]) and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0
    assert any(flips) and not all(flips)                  # the seed exercises both branches
    # deterministic per (seed, epoch, index); another epoch flips differently somewhere
    x2, _ = pipe.batch([0, 1, 2])
    assert torch.equal(x, x2)


def test_pipeline_feeds_a_training_step(tmp_path):
    """real decoded images through the whole path: pipeline -> ERDTrainer.train_step"""
    import e2e_util as U
    from erd_amd.datasets import CocoAnnotations, GpuDetPipeline
    from erd_amd.engine import ERDTrainer
    _make_dataset(tmp_path, [(120, 150), (140, 100)])
    ann = CocoAnnotations(str(tmp_path / "ann.json"), classes=("a", "b"), data_prefix=str(tmp_path), min_size=0)
    pipe = GpuDetPipeline(ann, scale=(160, 128), seed=1)
    x, samples = pipe.batch([0, 1])
    tsd, ssd = U.f7_state_dicts()
    tr = ERDTrainer(U.build_erd(tsd, ssd), lr=0.01, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
    log = tr.train_step(x, samples)
    tr.flush()
    assert np.isfinite(float(log["loss"])) and float(log["loss"]) > 0


def test_train_py_runs_on_real_files(tmp_path):
    """tools/train.py with the config's COCO files present: annotations -> shuffled aspect-ratio batches -> GPU pipeline ->
    trainer (2 iterations), through the CLI."""
    import subprocess, sys
    import e2e_util as U
    (tmp_path / "annotations").mkdir()
    (tmp_path / "train2017").mkdir()
    ds = _make_dataset(tmp_path / "train2017", [(120, 150), (140, 100), (100, 160), (150, 120)])
    for c, k in zip(ds["categories"], (1, 2)):
        c["name"] = f"cat{k}"
    json.dump(ds, open(tmp_path / "annotations" / "train.json", "w"))
    teacher = tmp_path / "teacher.pth"
    torch.save(dict(state_dict=O.procedural_state_dict(40, seed=0)), teacher)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "train.py"), U.CFG_INCRE, "--work-dir", str(tmp_path / "w"),
           "--max-iters", "2", "--cfg-options", "train_dataloader.batch_size=2",
           f"train_dataloader.dataset.data_root={tmp_path}/", "train_dataloader.dataset.ann_file=annotations/train.json",
           "train_dataloader.dataset.filter_cfg.min_size=0",
           f"model.ori_setting.ori_checkpoint_file={teacher}", f"model.ori_setting.ori_config_file={U.CFG_FIRST}",
           "default_hooks.logger.interval=1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Epoch(train) [1][2/2]" in out.stdout and "loss_dist_bbox" in out.stdout


def test_test_py_evaluates_a_checkpoint(tmp_path):
    """tools/test.py: checkpoint + val images -> predict (rescaled to the original image) -> COCO mAP table with the
    old / new split.  Procedural weights, so the numbers are ~0; the plumbing (ids, label->category, rescale) is what runs."""
    import subprocess, sys
    import e2e_util as U
    from erd_amd.runner import save_checkpoint
    (tmp_path / "val").mkdir()
    ds = _make_dataset(tmp_path / "val", [(120, 150), (140, 100), (100, 160)])
    ds["categories"] = [dict(id=10 + k, name=f"k{k}") for k in range(80)]
    for a in ds["annotations"]:
        a["category_id"] = 10 + 40 + a["category_id"]
    json.dump(ds, open(tmp_path / "val.json", "w"))
    tsd, ssd = U.f7_state_dicts()
    ckpt = tmp_path / "epoch_12.pth"
    save_checkpoint(str(ckpt), U.build_erd(tsd, ssd), with_teacher=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "test.py"), U.CFG_INCRE, str(ckpt), "--batch-size", "2", "--out",
           str(tmp_path / "res.json"), "--cfg-options", f"test_dataloader.dataset.data_root={tmp_path}/",
           "test_dataloader.dataset.ann_file=val.json", "test_dataloader.dataset.data_prefix.img=val/",
           "model.test_cfg.score_thr=0.001"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "bbox_mAP" in out.stdout and "old_mAP" in out.stdout and "new_mAP" in out.stdout
    res = json.load(open(tmp_path / "res.json"))
    assert set(res["stats"]) >= {"bbox_mAP", "bbox_mAP_50", "AR@100", "old_mAP", "new_mAP"} and len(res["classwise"]) == 80
    for r in res["results"]:                                   # boxes live in ORIGINAL image coordinates
        im = next(i for i in ds["images"] if i["id"] == r["image_id"])
        assert r["bbox"][0] >= -1e-3 and r["bbox"][0] + r["bbox"][2] <= im["width"] + 1e-2
        assert 10 <= r["category_id"] < 90


def test_threaded_decode_prefetch_gives_the_same_batches(tmp_path):
    """train_dataloader.num_workers decoding threads (pinned staging, two batches ahead) == in-thread decoding, bit for
    bit, in the same order"""
    from erd_amd.runner import CocoTrainData
    _make_dataset(tmp_path, [(48, 64), (60, 45), (33, 80), (50, 70), (64, 48), (40, 90), (70, 52)])
    dcfg = dict(data_root=str(tmp_path), ann_file="ann.json", data_prefix=dict(img=""), metainfo=dict(classes=("a", "b")),
                filter_cfg=dict(filter_empty_gt=True, min_size=0))
    runs = []
    for workers in (0, 3):
        data = CocoTrainData(dcfg, batch_size=2, scale=(133, 80), seed=5, num_workers=workers, prefetch_factor=2)
        data.set_epoch(1)
        runs.append([(b["inputs"].cpu(), [s.gt_instances.bboxes.cpu() for s in b["data_samples"]],
                      [s.metainfo["img_id"] for s in b["data_samples"]]) for b in data])
    assert len(runs[0]) == len(runs[1]) == 4
    for (xa, ba, ia), (xb, bb, ib) in zip(*runs):
        assert ia == ib and torch.equal(xa, xb) and all(torch.equal(p, q) for p, q in zip(ba, bb))
