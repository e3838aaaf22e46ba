"""Annotation side of the data pipeline: COCO json -> per-image records restricted to the
configured classes, the empty/min-size filter, aspect-ratio batching, category slicing for the 40+40 protocol.
Host-side integer / dictionary work only (the reference does this in Python too); pixels are decoded and resized
by `GpuDetPipeline` at the end of this file (decode on the host with PIL, everything else in ONE HIP kernel).

Reference: mmdet/datasets/coco.py:59-100 (load_data_list), :102-170 (parse_data_info), :172-212 (filter_data);
mmdet/datasets/samplers/batch_sampler.py:11-68; scripts/select_categories.py:21-64; pycocotools' COCO index
(getCatIds / getImgIds / getAnnIds semantics restated: dataset order everywhere).
"""
from __future__ import annotations

import json
import math
import os
from collections import defaultdict
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .structures import DetDataSample, InstanceData


def select_categories(dataset: dict, start: int, end: int) -> dict:
    """scripts/select_categories.py:31-60: sort categories by id, keep those at positions [start, end), then the
    annotations of those categories and the images that still have at least one annotation (original order kept)."""
    cats = sorted(dataset["categories"], key=lambda c: c["id"])[start:end]
    ids = {c["id"] for c in cats}
    annos = [a for a in dataset["annotations"] if a["category_id"] in ids]
    img_ids = {a["image_id"] for a in annos}
    return dict(categories=cats, annotations=annos, images=[im for im in dataset["images"] if im["id"] in img_ids])


class CocoAnnotations:
    """what the reference keeps of a COCO annotation file after load_data_list + filter_data"""

    def __init__(self, ann_file, classes: Sequence[str], data_prefix: str = "", filter_empty_gt: bool = True,
                 min_size: int = 32, test_mode: bool = False):
        ds = json.load(open(ann_file)) if isinstance(ann_file, (str, os.PathLike)) else ann_file
        if classes is None:      # CocoDataset's default METAINFO lists all 80 names: every category of the file matches
            classes = [c["name"] for c in ds["categories"]]
        names = set(classes)
        # getCatIds(catNms=classes): ids in the FILE's category order, not in `classes` order (coco.py:69-72)
        self.cat_ids = [c["id"] for c in ds["categories"] if c["name"] in names]
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        self.classes = tuple(classes)
        anns_of = defaultdict(list)
        cat_img_map = defaultdict(list)
        seen = set()
        for a in ds["annotations"]:
            if a["id"] in seen:
                raise AssertionError(f"Annotation ids in '{ann_file}' are not unique!")
            seen.add(a["id"])
            anns_of[a["image_id"]].append(a)
            cat_img_map[a["category_id"]].append(a["image_id"])
        self.data_list = [self._parse(im, anns_of.get(im["id"], []), data_prefix) for im in ds["images"]]
        if not test_mode:
            in_cat = set()
            for cid in self.cat_ids:
                in_cat |= set(cat_img_map.get(cid, []))
            self.data_list = [d for d in self.data_list
                              if not (filter_empty_gt and d["img_id"] not in in_cat) and
                              min(d["width"], d["height"]) >= min_size]

    def _parse(self, img: dict, anns: Iterable[dict], prefix: str) -> dict:
        W, H = img["width"], img["height"]
        inst = []
        for a in anns:
            if a.get("ignore", False):
                continue
            x1, y1, w, h = a["bbox"]
            iw = max(0, min(x1 + w, W) - max(x1, 0))
            ih = max(0, min(y1 + h, H) - max(y1, 0))
            if iw * ih == 0 or a["area"] <= 0 or w < 1 or h < 1 or a["category_id"] not in self.cat2label:
                continue
            inst.append(dict(bbox=[x1, y1, x1 + w, y1 + h], bbox_label=self.cat2label[a["category_id"]],
                             ignore_flag=1 if a.get("iscrowd", False) else 0))
        return dict(img_path=os.path.join(prefix, img["file_name"]), img_id=img["id"], height=H, width=W, instances=inst)

    def __len__(self):
        return len(self.data_list)

    def get_data_info(self, idx: int) -> dict:
        return self.data_list[idx]

    def data_sample(self, idx: int, scale_factor=(1.0, 1.0), flip: bool = False, img_shape=None, clip: bool = False) -> DetDataSample:
        """PackDetInputs for the annotation half: boxes of non-ignored instances scaled by (w_scale, h_scale) and
        optionally flipped horizontally inside img_shape (RandomFlip), ignored ones in `ignored_instances`."""
        d = self.data_list[idx]
        sw, sh = scale_factor
        shape = img_shape or (int(d["height"] * sh + 0.5), int(d["width"] * sw + 0.5))
        boxes = torch.tensor([i["bbox"] for i in d["instances"]], dtype=torch.float32).reshape(-1, 4)
        boxes = boxes * torch.tensor([sw, sh, sw, sh], dtype=torch.float32)
        if clip:           # Resize(clip_object_border=True): boxes clipped to the resized image
            boxes[:, 0::2].clamp_(0, shape[1])
            boxes[:, 1::2].clamp_(0, shape[0])
        if flip:
            x1 = shape[1] - boxes[:, 2]
            x2 = shape[1] - boxes[:, 0]
            boxes = torch.stack([x1, boxes[:, 1], x2, boxes[:, 3]], 1)
        labels = torch.tensor([i["bbox_label"] for i in d["instances"]], dtype=torch.int64)
        ign = torch.tensor([i["ignore_flag"] for i in d["instances"]], dtype=torch.bool)
        s = DetDataSample(metainfo=dict(img_id=d["img_id"], img_path=d["img_path"], ori_shape=(d["height"], d["width"]),
                                        img_shape=shape, scale_factor=(sw, sh), flip=flip))
        s.gt_instances = InstanceData(bboxes=boxes[~ign], labels=labels[~ign])
        s.ignored_instances = InstanceData(bboxes=boxes[ign], labels=labels[ign])
        return s


def rescale_size(old_wh, scale) -> tuple:
    """mmcv.image.rescale_size for a (long, short) target such as (1333, 800) with keep_ratio: the largest factor
    that keeps the long edge <= max(scale) and the short edge <= min(scale); new size = int(x * f + 0.5)."""
    w, h = old_wh
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


class AspectRatioBatchSampler:
    """batch_sampler.py:11-68: indices from `sampler` are bucketed by (width < height); a bucket is emitted when it
    reaches batch_size; leftovers of both buckets are concatenated (portrait first) and chunked at the end."""

    def __init__(self, sampler: Iterable[int], dataset: CocoAnnotations, batch_size: int, drop_last: bool = False):
        if not isinstance(batch_size, int) or batch_size <= 0:
            raise ValueError(f"batch_size should be a positive integer value, but got batch_size={batch_size}")
        self.sampler, self.dataset, self.batch_size, self.drop_last = sampler, dataset, batch_size, drop_last

    def __iter__(self) -> Iterator[List[int]]:
        buckets = [[], []]
        for idx in self.sampler:
            d = self.dataset.get_data_info(idx)
            b = buckets[0 if d["width"] < d["height"] else 1]
            b.append(idx)
            if len(b) == self.batch_size:
                yield b[:]
                del b[:]
        left = buckets[0] + buckets[1]
        while left:
            if len(left) <= self.batch_size:
                if not self.drop_last:
                    yield left[:]
                left = []
            else:
                yield left[:self.batch_size]
                left = left[self.batch_size:]

    def __len__(self) -> int:
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


# ---------------------------------------------------------------------------------------------------------
# image side: LoadImageFromFile -> Resize(scale=(1333, 800), keep_ratio=True) -> RandomFlip(0.5) -> PackDetInputs ->
# DetDataPreprocessor (configs/gfl_increment/*:13-19, data_preprocessor.py:110-183).  Decode stays on the host (PIL; the
# reference decodes with cv2 on the host too); resize + flip + normalise + pad are one kernel over the padded batch slot
# (erd_resize_normalize).  cv2's 8-bit bilinear resize is restated (resize.cpp: half-pixel centres, 11-bit fixed-point
# weights, two-pass rounding) -- UNPINNED against cv2, which is not in this image.
# ---------------------------------------------------------------------------------------------------------
def linear_coeffs(src: int, dst: int):
    """per output index along one axis: source index (int32) and the two fixed-point weights (int16, sum 2048)"""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= src - 1
    f[lo], s[lo] = 0.0, 0
    f[hi], s[hi] = 0.0, src - 1
    c1 = np.rint(f.astype(np.float64) * 2048.0)
    c0 = np.rint((1.0 - f).astype(np.float32).astype(np.float64) * 2048.0)
    return s, np.stack([c0, c1], 1).astype(np.int16)


def load_image_bgr(path: str) -> np.ndarray:
    """LoadImageFromFile: uint8 [h, w, 3] in BGR order (what cv2.imread hands the reference's pipeline)"""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def prefetch_map(fn, items: Sequence, workers: int, depth: int):
    """yield fn(item) for every item IN ORDER while up to `depth` later items are already being computed on `workers`
    threads (the DataLoader's num_workers / prefetch_factor, as threads: image decoding releases the GIL).  workers <= 0:
    plain serial map.  An exception in fn surfaces at the position of its item."""
    if workers <= 0:
        for it in items:
            yield fn(it)
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="erd-decode") as pool:
        pending = deque()
        it = iter(items)
        try:
            for _ in range(max(1, depth)):
                pending.append(pool.submit(fn, next(it)))
        except StopIteration:
            pass
        while pending:
            head = pending.popleft()
            try:
                pending.append(pool.submit(fn, next(it)))
            except StopIteration:
                pass
            try:
                yield head.result()
            except BaseException:
                for f in pending:
                    f.cancel()
                raise


def pinned(im: np.ndarray):
    """page-locked copy of a decoded image when a GPU is present (so that the H2D copy is asynchronous)"""
    t = torch.from_numpy(im)
    return t.pin_memory() if torch.cuda.is_available() else t


class GpuDetPipeline:
    """one training batch from image indices: decoded images -> normalised, padded [N,3,H,W] fp32 on the GPU + data
    samples with resized / flipped / clipped boxes.  Deterministic given `seed` (flip decisions per (epoch, index))."""

    def __init__(self, annotations: CocoAnnotations, scale=(1333, 800), flip_prob: float = 0.5, mean=(123.675, 116.28, 103.53),
                 std=(58.395, 57.12, 57.375), bgr_to_rgb: bool = True, pad_size_divisor: int = 32, pad_value: float = 0.0,
                 seed: int = 0, loader=load_image_bgr, device="cuda"):
        self.ann, self.scale, self.flip_prob = annotations, tuple(scale), flip_prob
        self.mean = [float(np.float32(v)) for v in mean]
        self.std = [float(np.float32(v)) for v in std]
        self.swap, self.div, self.pad_value, self.seed = bgr_to_rgb, pad_size_divisor, pad_value, seed
        self.loader, self.device = loader, torch.device(device)
        self._tables: Dict[tuple, tuple] = {}
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _table(self, sh, sw, nh, nw):
        key = (sh, sw, nh, nw)
        if key not in self._tables:
            xo, xc = linear_coeffs(sw, nw)
            yo, yc = linear_coeffs(sh, nh)
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device) for a in (xo, xc, yo, yc))
        return self._tables[key]

    def decode(self, indices: Sequence[int]) -> List[np.ndarray]:
        """host half of a batch (LoadImageFromFile): safe to run on worker threads, PIL releases the GIL while decoding"""
        return [self.loader(self.ann.get_data_info(i)["img_path"]) for i in indices]

    def batch(self, indices: Sequence[int]):
        return self.assemble(indices, self.decode(indices))

    def assemble(self, indices: Sequence[int], imgs: Sequence[np.ndarray]):
        """device half: resize / flip / normalise / pad kernels + the data samples"""
        from . import kernels as K
        new = [rescale_size((im.shape[1], im.shape[0]), self.scale) for im in imgs]          # (w, h)
        H = max(int(math.ceil(h / self.div)) * self.div for _, h in new)
        W = max(int(math.ceil(w / self.div)) * self.div for w, _ in new)
        out = torch.empty((len(imgs), 3, H, W), dtype=torch.float32, device=self.device)
        samples = []
        for k, (i, im, (nw, nh)) in enumerate(zip(indices, imgs, new)):
            rng = np.random.RandomState((self.seed * 1000003 + self.epoch * 7919 + int(i)) % (2 ** 31 - 1))
            flip = bool(rng.rand() < self.flip_prob)
            src = (im if isinstance(im, torch.Tensor) else torch.from_numpy(im)).to(self.device, non_blocking=True)
            K.resize_normalize_into(src, self._table(im.shape[0], im.shape[1], nh, nw), (nh, nw), out[k], self.mean, self.std,
                                    flip, self.swap, self.pad_value)
            s = self.ann.data_sample(i, scale_factor=(nw / im.shape[1], nh / im.shape[0]), flip=flip, img_shape=(nh, nw),
                                     clip=True)
            s.set_metainfo(dict(pad_shape=(H, W), batch_input_shape=(H, W)))
            samples.append(s)
        return out, samples
