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
