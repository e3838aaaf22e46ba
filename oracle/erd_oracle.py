"""TEST INFRASTRUCTURE -- CPU restatement ("port") of the ERD incremental training step.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``erd_amd``) never does, and fails loudly without its
HIP library.

It restates, in plain functional PyTorch-CPU fp32 (no mmengine / mmcv / mmdet),
the hot path of Hi-FT/ERD (`/root/reference`, a fork of MMDetection 3.0.0):
teacher forward -> Elastic Response Selection -> student forward -> new-class
QFL/GIoU/DFL -> L2 + NMS-filtered KL response distillation.  Every function
cites the reference file:line it follows.

PARITY PIN: the reference's own tests never touch the ERD path (SURVEY.md
section 4), so the pin is (a) the known-answer vectors the reference does hold
for pieces of the path (ATSS 4x2 case, anchor grids, valid-flag counts,
weighted-loss docstring values -- ``tests/test_oracle_known_answers.py``) and
(b) outputs of the reference's own source executed in the build container via
``oracle/ref_stub.py`` (fixtures ``tests/golden/*.npz`` written by
``oracle/gen_golden.py``; live comparison in ``tests/test_oracle_vs_reference.py``
whenever ``/root/reference`` is present).  ``mmcv.ops.batched_nms`` is absent
from the tree and the image: its restatement is UNPINNED vs. mmcv==2.0.0.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS32 = float(torch.finfo(torch.float32).eps)

# ----------------------------------------------------------------------------
# architecture tables  (resnet.py:361-367 arch_settings; Bottleneck expansion 4)
# ----------------------------------------------------------------------------
RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
STRIDES = (8, 16, 32, 64, 128)
REG_MAX = 16


def resnet_param_shapes(depth: int = 50) -> Dict[str, Tuple[int, ...]]:
    """state-dict keys/shapes of `backbone.*` (resnet.py:565-611,:97-262; res_layer.py:31-109)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def bn(prefix, c):
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)
        s[prefix + ".running_mean"] = (c,)
        s[prefix + ".running_var"] = (c,)
        s[prefix + ".num_batches_tracked"] = ()

    s["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, nblk in enumerate(RESNET_BLOCKS[depth]):
        planes = 64 * 2 ** li
        for b in range(nblk):
            p = f"layer{li + 1}.{b}"
            s[p + ".conv1.weight"] = (planes, inplanes, 1, 1)
            bn(p + ".bn1", planes)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3)
            bn(p + ".bn2", planes)
            s[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(p + ".bn3", planes * 4)
            if b == 0:
                s[p + ".downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    return s


def gfl_param_shapes(num_classes: int, depth: int = 50) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint ABI of a GFL detector (SURVEY.md 8(b) state-dict keys)."""
    s = {"backbone." + k: v for k, v in resnet_param_shapes(depth).items()}
    for i, c in enumerate((512, 1024, 2048)):
        s[f"neck.lateral_convs.{i}.conv.weight"] = (256, c, 1, 1)
        s[f"neck.lateral_convs.{i}.conv.bias"] = (256,)
    for i in range(5):
        s[f"neck.fpn_convs.{i}.conv.weight"] = (256, 256, 3, 3)
        s[f"neck.fpn_convs.{i}.conv.bias"] = (256,)
    for tower in ("cls_convs", "reg_convs"):
        for i in range(4):
            s[f"bbox_head.{tower}.{i}.conv.weight"] = (256, 256, 3, 3)
            s[f"bbox_head.{tower}.{i}.gn.weight"] = (256,)
            s[f"bbox_head.{tower}.{i}.gn.bias"] = (256,)
    s["bbox_head.gfl_cls.weight"] = (num_classes, 256, 3, 3)
    s["bbox_head.gfl_cls.bias"] = (num_classes,)
    s["bbox_head.gfl_reg.weight"] = (4 * (REG_MAX + 1), 256, 3, 3)
    s["bbox_head.gfl_reg.bias"] = (4 * (REG_MAX + 1),)
    for i in range(5):
        s[f"bbox_head.scales.{i}.scale"] = ()
    s["bbox_head.integral.project"] = (REG_MAX + 1,)
    return s


def _name_seed(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def procedural_tensor(seed: int, name: str, shape: Tuple[int, ...]) -> Tensor:
    """w[name] = f(seed, name, shape): deterministic, platform-independent (numpy
    PCG64 -> float32), scaled so that activations stay O(1) through ~50 layers.
    This is the *synthetic-weight spec* shared by the oracle, the product, the
    fixtures and bench.py (no 130 MB checkpoints are committed)."""
    rng = np.random.Generator(np.random.PCG64(_name_seed(seed, name)))
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "project":
        return torch.linspace(0, REG_MAX, REG_MAX + 1)
    if leaf == "scale":
        return torch.tensor(float(0.9 + 0.2 * rng.random()), dtype=torch.float32)
    if leaf == "running_mean":
        return torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
    if leaf == "running_var":
        return torch.from_numpy((0.5 + rng.random(shape)).astype(np.float32))
    is_norm = (".bn" in name or ".gn." in name or "downsample.1" in name or name.startswith("backbone.bn1")
               or ".bn1." in name)
    if leaf == "weight" and len(shape) == 1:
        lo = 0.25 if (".bn3." in name) else 0.75      # damp residual-branch growth
        return torch.from_numpy((lo + 0.5 * rng.random(shape)).astype(np.float32))
    if leaf == "bias" and is_norm:
        return torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
    if leaf == "bias":
        if "gfl_cls" in name:      # bias_prob=0.01 -> -log(99) (gfl_head_increment_erd.py:109-117)
            return torch.full(shape, -4.59511985013459, dtype=torch.float32)
        return torch.from_numpy((0.05 * rng.standard_normal(shape)).astype(np.float32))
    if leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        std = math.sqrt(2.0 / fan_in)
        if "gfl_cls" in name:
            std = 0.5 * std
        if "gfl_reg" in name:
            std = 1.5 * std
        return torch.from_numpy((std * rng.standard_normal(shape)).astype(np.float32))
    raise KeyError(name)


def procedural_state_dict(num_classes: int, depth: int = 50, seed: int = 0,
                          prefix: str = "") -> Dict[str, Tensor]:
    return {prefix + k: procedural_tensor(seed, k, shp)
            for k, shp in gfl_param_shapes(num_classes, depth).items()}


def student_state_from_teacher(teacher_sd: Dict[str, Tensor], num_classes: int, seed: int = 1,
                               depth: int = 50) -> Dict[str, Tensor]:
    """gfl_increment_erd.py:67-93: student = teacher checkpoint with `gfl_cls`
    widened 40->80 rows (old rows copied, new rows = the student's own fresh init)."""
    sd = {k: v.clone() for k, v in teacher_sd.items()}
    w_new = procedural_tensor(seed, "bbox_head.gfl_cls.weight", (num_classes, 256, 3, 3))
    b_new = procedural_tensor(seed, "bbox_head.gfl_cls.bias", (num_classes,))
    c_old = teacher_sd["bbox_head.gfl_cls.weight"].shape[0]
    sd["bbox_head.gfl_cls.weight"] = torch.cat([teacher_sd["bbox_head.gfl_cls.weight"], w_new[c_old:]], 0)
    sd["bbox_head.gfl_cls.bias"] = torch.cat([teacher_sd["bbox_head.gfl_cls.bias"], b_new[c_old:]], 0)
    return sd


# ----------------------------------------------------------------------------
# network forward (functional)
# ----------------------------------------------------------------------------
# BASELINE.json configs[2] ("bf16"): the HIP path can run every 1x1 / 3x3 convolution on the bf16 matrix cores --
# both multiplicands rounded to bf16 (round-to-nearest-even), products exact, accumulation / normalisation /
# losses / storage in fp32; the 7x7 stem stays fp32.  `with bf16_multiplicands():` makes the restatement do the
# same, so the forward pass can be compared as tightly as in fp32 (only the summation order differs).
_BF16_MULTIPLICANDS = False


class bf16_multiplicands:
    def __enter__(self):
        global _BF16_MULTIPLICANDS
        self.prev, _BF16_MULTIPLICANDS = _BF16_MULTIPLICANDS, True

    def __exit__(self, *exc):
        global _BF16_MULTIPLICANDS
        _BF16_MULTIPLICANDS = self.prev


# `with bf16_stored_maps():` additionally rounds every STORED feature map to bf16 where the HIP path's bf16 mode stores one (DESIGN 8.0c:
# maps and their gradients live in HBM as bf16, what the reference's AMP switch does to conv outputs, tools/train.py:85-97) -- the stem's
# and every conv + BN (+ ReLU) block's output, the residual sum behind its ReLU, the projection shortcut, FPN laterals / top-down sums /
# outputs, the raw tower convolutions (the GroupNorm INPUT) and the GroupNorm + ReLU outputs; head outputs, statistics, losses and
# parameters stay fp32.  The gradient that flows back through a stored map is rounded too (gradient maps are bf16 maps).  With the
# rounding points shared, the discrete decisions of the step (ReLU masks, ERS thresholds, ATSS candidates) fall the same way on both
# sides, which "multiplicands only" cannot promise: tests/test_gpu_e2e.py holds the HIP bf16 step to a FRACTION of the bf16-vs-fp32 noise.
_BF16_STORED = False


class bf16_stored_maps:
    def __enter__(self):
        global _BF16_MULTIPLICANDS, _BF16_STORED
        self.prev = (_BF16_MULTIPLICANDS, _BF16_STORED)
        _BF16_MULTIPLICANDS = _BF16_STORED = True

    def __exit__(self, *exc):
        global _BF16_MULTIPLICANDS, _BF16_STORED
        _BF16_MULTIPLICANDS, _BF16_STORED = self.prev


class _StoreBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _store(x):
    """a feature map as the bf16 mode keeps it in memory (identity outside `bf16_stored_maps`)"""
    return _StoreBF16.apply(x) if _BF16_STORED else x


def _conv2d(x, w, b=None, stride=1, padding=0):
    if _BF16_MULTIPLICANDS:
        x, w = x.to(torch.bfloat16).to(torch.float32), w.to(torch.bfloat16).to(torch.float32)
    return F.conv2d(x, w, b, stride, padding)


def _bn_eval(x, sd, p):
    """Frozen-statistics BN: every BN runs in eval mode (norm_eval=True, resnet.py:648-657)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def resnet_stem(sub: Dict[str, Tensor], x: Tensor) -> Tensor:
    """resnet.py:631-640: conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 (`sub`: the backbone's state dict without its prefix)"""
    x = F.conv2d(x, sub["conv1.weight"], None, 2, 3)
    x = _store(F.relu(_bn_eval(x, sub, "bn1")))
    return F.max_pool2d(x, 3, 2, 1)


def resnet_block(sub: Dict[str, Tensor], x: Tensor, li: int, b: int) -> Tensor:
    """Bottleneck.forward, resnet.py:263-302 (style='pytorch': stride on conv2; block 0 of a stage carries the projection shortcut)"""
    p = f"layer{li + 1}.{b}"
    stride = 2 if (b == 0 and li > 0) else 1
    identity = x
    out = _store(F.relu(_bn_eval(_conv2d(x, sub[p + ".conv1.weight"]), sub, p + ".bn1")))
    out = _store(F.relu(_bn_eval(_conv2d(out, sub[p + ".conv2.weight"], None, stride, 1), sub, p + ".bn2")))
    out = _bn_eval(_conv2d(out, sub[p + ".conv3.weight"]), sub, p + ".bn3")
    if b == 0:
        identity = _store(_bn_eval(_conv2d(x, sub[p + ".downsample.0.weight"], None, stride), sub, p + ".downsample.1"))
    return _store(F.relu(out + identity))


def resnet_layer(sub: Dict[str, Tensor], x: Tensor, li: int, depth: int = 50) -> Tensor:
    """one stage (`layer{li+1}`) of Bottleneck blocks, res_layer.py:57-63"""
    for b in range(RESNET_BLOCKS[depth][li]):
        x = resnet_block(sub, x, li, b)
    return x


def resnet_forward(sd: Dict[str, Tensor], x: Tensor, depth: int = 50, prefix: str = "backbone.") -> List[Tensor]:
    """resnet.py:631-646 + Bottleneck.forward :263-302."""
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = resnet_stem(sub, x)
    outs = []
    for li in range(len(RESNET_BLOCKS[depth])):
        x = resnet_layer(sub, x, li, depth)
        outs.append(x)
    return outs


def fpn_forward(sd: Dict[str, Tensor], feats: Sequence[Tensor], prefix: str = "neck.") -> List[Tensor]:
    """fpn.py:161-221 with start_level=1, add_extra_convs='on_output', num_outs=5,
    no norm / no activation, nearest top-down with size= (fpn.py:181-191)."""
    g = lambda k: sd[prefix + k]
    ins = feats[1:]
    lats = [_store(_conv2d(ins[i], g(f"lateral_convs.{i}.conv.weight"), g(f"lateral_convs.{i}.conv.bias")))
            for i in range(3)]
    for i in range(2, 0, -1):
        lats[i - 1] = _store(lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode="nearest"))
    outs = [_store(_conv2d(lats[i], g(f"fpn_convs.{i}.conv.weight"), g(f"fpn_convs.{i}.conv.bias"), 1, 1))
            for i in range(3)]
    outs.append(_store(_conv2d(outs[-1], g("fpn_convs.3.conv.weight"), g("fpn_convs.3.conv.bias"), 2, 1)))
    outs.append(_store(_conv2d(outs[-1], g("fpn_convs.4.conv.weight"), g("fpn_convs.4.conv.bias"), 2, 1)))
    return outs


def gfl_head_forward(sd: Dict[str, Tensor], feats: Sequence[Tensor], prefix: str = "bbox_head.") \
        -> Tuple[List[Tensor], List[Tensor]]:
    """gfl_head.py:186-230: 4x(conv3x3 no-bias -> GN(32) -> ReLU) per tower (weights shared
    over levels), gfl_cls, gfl_reg * Scale_l, .float()."""
    g = lambda k: sd[prefix + k]
    cls_scores, bbox_preds = [], []
    for l, x in enumerate(feats):
        c, r = x, x
        for i in range(4):
            c = head_tower_layer(sd, c, "cls", i, prefix)
            r = head_tower_layer(sd, r, "reg", i, prefix)
        cls_scores.append(_conv2d(c, g("gfl_cls.weight"), g("gfl_cls.bias"), 1, 1))
        bbox_preds.append((_conv2d(r, g("gfl_reg.weight"), g("gfl_reg.bias"), 1, 1)
                           * g(f"scales.{l}.scale")).float())
    return cls_scores, bbox_preds


def head_tower_layer(sd: Dict[str, Tensor], x: Tensor, branch: str, i: int, prefix: str = "bbox_head.") -> Tensor:
    """one ConvModule of a tower on one level: conv3x3 (no bias) -> GroupNorm(32) -> ReLU, gfl_head.py:158-177, 219-222"""
    g = lambda k: sd[prefix + k]
    return _store(F.relu(F.group_norm(_store(_conv2d(x, g(f"{branch}_convs.{i}.conv.weight"), None, 1, 1)), 32,
                                      g(f"{branch}_convs.{i}.gn.weight"), g(f"{branch}_convs.{i}.gn.bias"), 1e-5)))


def gfl_forward(sd: Dict[str, Tensor], x: Tensor, depth: int = 50) -> Tuple[List[Tensor], List[Tensor]]:
    """single_stage.py:116-149 `_forward` (mode='tensor')."""
    return gfl_head_forward(sd, fpn_forward(sd, resnet_forward(sd, x, depth)))


def flatten_levels(maps: Sequence[Tensor], channels: Optional[slice] = None) -> Tensor:
    """permute NCHW->NHWC, reshape [N,hw,C], cat levels (gfl_increment_erd.py:183-193)."""
    out = []
    for m in maps:
        if channels is not None:
            m = m[:, channels]
        out.append(m.permute(0, 2, 3, 1).reshape(m.shape[0], -1, m.shape[1]))
    return torch.cat(out, 1)


# ----------------------------------------------------------------------------
# data preprocessor (data_preprocessor.py:110-183 + mmengine ImgDataPreprocessor)
# ----------------------------------------------------------------------------
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


def preprocess(images: Sequence[Tensor], divisor: int = 32, bgr_to_rgb: bool = True) \
        -> Tuple[Tensor, List[dict]]:
    """uint8 [3,h,w] BGR images -> fp32 [N,3,H32,W32]: channel flip, float, (x-mean)/std,
    zero-pad bottom/right AFTER normalisation; stamps pad_shape / batch_input_shape."""
    mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(3, 1, 1)
    H = max(int(math.ceil(im.shape[1] / divisor)) * divisor for im in images)
    W = max(int(math.ceil(im.shape[2] / divisor)) * divisor for im in images)
    batch = torch.zeros(len(images), 3, H, W)
    metas = []
    for i, im in enumerate(images):
        x = im[[2, 1, 0]] if bgr_to_rgb else im
        x = (x.float() - mean) / std
        batch[i, :, :x.shape[1], :x.shape[2]] = x
        metas.append(dict(img_shape=(im.shape[1], im.shape[2]), pad_shape=(H, W),
                          batch_input_shape=(H, W)))
    return batch, metas


# ----------------------------------------------------------------------------
# ERS -- Elastic Response Selection (gfl_increment_erd.py:143-163)
# ----------------------------------------------------------------------------
def ers_select_single(cls_scores: Tensor, bbox_preds: Tensor) -> Tuple[Tensor, Tensor, float, float]:
    """cls: keep a iff max_k sigmoid(cls[a,k]) > mean + 2*std (unbiased) over ALL anchors;
    bbox: keep a iff max_j bbox_logit[a,j] (raw, all 68) > mean + 2*std.  Strict '>'."""
    m_c = cls_scores.sigmoid().max(dim=-1)[0]
    thr_c = m_c.mean() + 2 * m_c.std()
    idx_c = (m_c > thr_c).nonzero(as_tuple=False).squeeze(1)
    m_b = bbox_preds.max(dim=-1)[0]
    thr_b = m_b.mean() + 2 * m_b.std()
    idx_b = (m_b > thr_b).nonzero(as_tuple=False).squeeze(1)
    return idx_c, idx_b, float(thr_c), float(thr_b)


# ----------------------------------------------------------------------------
# anchors (anchor_generator.py:161-205, 259-301, 415-476; anchor_head.py:164-199)
# ----------------------------------------------------------------------------
def grid_anchors(featmap_sizes: Sequence[Tuple[int, int]], strides: Sequence[int] = STRIDES,
                 octave_base_scale: int = 8) -> List[Tensor]:
    """square anchors of side 8*stride centred on (x*s, y*s) (center_offset 0), row-major."""
    out = []
    for (h, w), s in zip(featmap_sizes, strides):
        half = 0.5 * (s * octave_base_scale)
        base = torch.tensor([-half, -half, half, half], dtype=torch.float32)
        sx = torch.arange(0, w).float() * s
        sy = torch.arange(0, h).float() * s
        xx = sx.repeat(h)
        yy = sy.view(-1, 1).repeat(1, w).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], -1)
        out.append(shifts + base[None])
    return out


def valid_flags(featmap_sizes, pad_shape, strides: Sequence[int] = STRIDES) -> List[Tensor]:
    out = []
    for (fh, fw), s in zip(featmap_sizes, strides):
        h, w = pad_shape[:2]
        vh = min(int(np.ceil(h / s)), fh)
        vw = min(int(np.ceil(w / s)), fw)
        vx = torch.zeros(fw, dtype=torch.bool)
        vy = torch.zeros(fh, dtype=torch.bool)
        vx[:vw] = True
        vy[:vh] = True
        out.append((vy[:, None] & vx[None, :]).reshape(-1))
    return out


def anchor_centers(anchors: Tensor) -> Tensor:
    """gfl_head.py:232-243."""
    return torch.stack([(anchors[..., 2] + anchors[..., 0]) / 2,
                        (anchors[..., 3] + anchors[..., 1]) / 2], -1)


# ----------------------------------------------------------------------------
# boxes (bbox_overlaps.py:13-199; transforms.py:147-230)
# ----------------------------------------------------------------------------
def bbox_overlaps(b1: Tensor, b2: Tensor, mode: str = "iou", is_aligned: bool = False,
                  eps: float = 1e-6) -> Tensor:
    rows, cols = b1.shape[-2], b2.shape[-2]
    if rows * cols == 0:
        return b1.new_zeros((rows,) if is_aligned else (rows, cols))
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if is_aligned:
        lt = torch.max(b1[..., :2], b2[..., :2])
        rb = torch.min(b1[..., 2:], b2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1 + a2 - overlap
        if mode == "giou":
            elt = torch.min(b1[..., :2], b2[..., :2])
            erb = torch.max(b1[..., 2:], b2[..., 2:])
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1[..., None] + a2[..., None, :] - overlap
        if mode == "giou":
            elt = torch.min(b1[..., :, None, :2], b2[..., None, :, :2])
            erb = torch.max(b1[..., :, None, 2:], b2[..., None, :, 2:])
    e = union.new_tensor([eps])
    union = torch.max(union, e)
    ious = overlap / union
    if mode == "iou":
        return ious
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], e)
    return ious - (earea - union) / earea


def distance2bbox(points: Tensor, distance: Tensor) -> Tensor:
    return torch.stack([points[..., 0] - distance[..., 0], points[..., 1] - distance[..., 1],
                        points[..., 0] + distance[..., 2], points[..., 1] + distance[..., 3]], -1)


def bbox2distance(points: Tensor, bbox: Tensor, max_dis: float, eps: float = 0.1) -> Tensor:
    l = (points[..., 0] - bbox[..., 0]).clamp(min=0, max=max_dis - eps)
    t = (points[..., 1] - bbox[..., 1]).clamp(min=0, max=max_dis - eps)
    r = (bbox[..., 2] - points[..., 0]).clamp(min=0, max=max_dis - eps)
    b = (bbox[..., 3] - points[..., 1]).clamp(min=0, max=max_dis - eps)
    return torch.stack([l, t, r, b], -1)


def integral(x: Tensor, reg_max: int = REG_MAX) -> Tensor:
    """gfl_head.py:29-62: softmax over 17 bins, expectation against [0..16]."""
    p = F.softmax(x.reshape(-1, reg_max + 1), dim=1)
    return F.linear(p, torch.linspace(0, reg_max, reg_max + 1).type_as(p)).reshape(-1, 4)


# ----------------------------------------------------------------------------
# ATSS assignment (atss_assigner.py:74-254) + target plumbing (gfl_head.py:562-669)
# ----------------------------------------------------------------------------
def atss_assign(priors: Tensor, num_level_priors: Sequence[int], gt_bboxes: Tensor, gt_labels: Tensor,
                topk: int = 9) -> Tuple[Tensor, Tensor]:
    """returns (assigned_gt_inds [n] 0=bg / i+1, assigned_labels [n] -1=bg)."""
    INF = 100000000
    num_gt, n = gt_bboxes.shape[0], priors.shape[0]
    assigned = priors.new_zeros((n,), dtype=torch.long)
    labels = priors.new_full((n,), -1, dtype=torch.long)
    if num_gt == 0 or n == 0:
        return assigned, labels
    overlaps = bbox_overlaps(priors, gt_bboxes)
    gcx = (gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0
    gcy = (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0
    pcx = (priors[:, 0] + priors[:, 2]) / 2.0
    pcy = (priors[:, 1] + priors[:, 3]) / 2.0
    pp = torch.stack((pcx, pcy), 1)
    gp = torch.stack((gcx, gcy), 1)
    distances = (pp[:, None, :] - gp[None, :, :]).pow(2).sum(-1).sqrt()
    cand = []
    start = 0
    for npl in num_level_priors:
        end = start + npl
        k = min(topk, npl)
        # tie-break: smallest distance first, then LOWEST index (stable); the HIP kernel does the same.
        d = distances[start:end]
        order = torch.sort(d, dim=0, stable=True).indices[:k]
        cand.append(order + start)
        start = end
    cand = torch.cat(cand, 0)                                   # [sum k, G]
    cand_ov = overlaps[cand, torch.arange(num_gt)]
    thr = cand_ov.mean(0) + cand_ov.std(0)
    is_pos = cand_ov >= thr[None, :]
    l_ = pcx[cand] - gt_bboxes[:, 0]
    t_ = pcy[cand] - gt_bboxes[:, 1]
    r_ = gt_bboxes[:, 2] - pcx[cand]
    b_ = gt_bboxes[:, 3] - pcy[cand]
    is_in = torch.stack([l_, t_, r_, b_], 1).min(1)[0] > 0.01
    is_pos = is_pos & is_in
    ov_inf = torch.full((num_gt, n), float(-INF))
    gi = torch.arange(num_gt)[None, :].expand_as(cand)
    ov_inf[gi[is_pos], cand[is_pos]] = overlaps[cand[is_pos], gi[is_pos]]
    max_ov, argmax = ov_inf.t().max(1)
    hit = max_ov != -INF
    assigned[hit] = argmax[hit] + 1
    pos = assigned > 0
    labels[pos] = gt_labels[assigned[pos] - 1]
    return assigned, labels


def get_targets_single(flat_anchors: Tensor, flags: Tensor, num_level_anchors: Sequence[int],
                       gt_bboxes: Tensor, gt_labels: Tensor, num_classes: int):
    """gfl_head.py:562-669 with allowed_border=-1 (inside = valid flags), PseudoSampler,
    pos_weight=-1.  Returns labels [A] (bg = num_classes), label_weights [A], bbox_targets [A,4],
    num_pos (int)."""
    if not bool(flags.any()):
        raise ValueError("There is no valid anchor inside the image boundary.")
    anchors = flat_anchors[flags]
    nl_inside = [int(f.sum()) for f in torch.split(flags, list(num_level_anchors))]
    assigned, _ = atss_assign(anchors, nl_inside, gt_bboxes, gt_labels)
    nv = anchors.shape[0]
    bt = torch.zeros_like(anchors)
    lab = anchors.new_full((nv,), num_classes, dtype=torch.long)
    lw = anchors.new_zeros(nv)
    pos = (assigned > 0).nonzero(as_tuple=False).squeeze(-1).unique()
    neg = (assigned == 0).nonzero(as_tuple=False).squeeze(-1).unique()
    if pos.numel() > 0:
        bt[pos] = gt_bboxes[assigned[pos] - 1]
        lab[pos] = gt_labels[assigned[pos] - 1]
        lw[pos] = 1.0
    if neg.numel() > 0:
        lw[neg] = 1.0
    A = flat_anchors.shape[0]
    lab_full = flat_anchors.new_full((A,), num_classes, dtype=torch.long)
    lab_full[flags] = lab
    lw_full = flat_anchors.new_zeros(A)
    lw_full[flags] = lw
    bt_full = flat_anchors.new_zeros(A, 4)
    bt_full[flags] = bt
    return lab_full, lw_full, bt_full, int(pos.numel())


# ----------------------------------------------------------------------------
# leaf losses
# ----------------------------------------------------------------------------
def quality_focal_loss(pred: Tensor, label: Tensor, score: Tensor, beta: float = 2.0) -> Tensor:
    """gfocal_loss.py:12-53 (row losses [n])."""
    ps = pred.sigmoid()
    loss = F.binary_cross_entropy_with_logits(pred, torch.zeros_like(pred), reduction="none") * ps.pow(beta)
    bg = pred.shape[1]
    pos = ((label >= 0) & (label < bg)).nonzero().squeeze(1)
    pl = label[pos].long()
    sf = score[pos] - ps[pos, pl]
    loss[pos, pl] = F.binary_cross_entropy_with_logits(pred[pos, pl], score[pos], reduction="none") \
        * sf.abs().pow(beta)
    return loss.sum(dim=1)


def distribution_focal_loss(pred: Tensor, label: Tensor) -> Tensor:
    """gfocal_loss.py:143-165."""
    dl = label.long()
    dr = dl + 1
    wl = dr.float() - label
    wr = label - dl.float()
    return F.cross_entropy(pred, dl, reduction="none") * wl + F.cross_entropy(pred, dr, reduction="none") * wr


def kd_kl_div(pred: Tensor, soft_label: Tensor, T: float) -> Tensor:
    """kd_loss.py:12-37."""
    target = F.softmax(soft_label / T, dim=1).detach()
    return F.kl_div(F.log_softmax(pred / T, dim=1), target, reduction="none").mean(1) * (T * T)


def weight_reduce(loss: Tensor, weight: Optional[Tensor], avg_factor: Optional[float]) -> Tensor:
    """losses/utils.py:30-65 (reduction='mean')."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean()
    return loss.sum() / (avg_factor + EPS32)


def giou_loss_module(pred: Tensor, target: Tensor, weight: Tensor, avg_factor: float,
                     loss_weight: float = 2.0, eps: float = 1e-6) -> Tensor:
    """iou_loss.py:463-528 (early-out when no positive weight)."""
    if not bool(torch.any(weight > 0)):
        return (pred * weight.unsqueeze(1)).sum()
    loss = 1 - bbox_overlaps(pred, target, mode="giou", is_aligned=True, eps=eps)
    return loss_weight * weight_reduce(loss, weight, avg_factor)


def nms_class_offset(boxes: Tensor, scores: Tensor, ids: Tensor, iou_thr: float) -> Tensor:
    """mmcv.ops.batched_nms restated (UNPINNED vs mmcv==2.0.0): fp32 offsets
    id*(max_coord+1), greedy suppress IoU > thr, keep in score-descending (stable) order."""
    if boxes.shape[0] == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    off = ids.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes))
    b = boxes + off[:, None]
    order = torch.sort(scores, descending=True, stable=True).indices
    b = b[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    supp = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if supp[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            lt = torch.maximum(b[i, :2], b[i + 1:, :2])
            rb = torch.minimum(b[i, 2:], b[i + 1:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            iou = inter / (area[i] + area[i + 1:] - inter)
            supp[i + 1:] |= iou > iou_thr
    return order[torch.tensor(keep, dtype=torch.long)]


# ----------------------------------------------------------------------------
# head losses (gfl_head_increment_erd.py)
# ----------------------------------------------------------------------------
def loss_by_feat_single(anchors: Tensor, cls_score: Tensor, bbox_pred: Tensor, labels: Tensor,
                        label_weights: Tensor, bbox_targets: Tensor, stride: int, c_old: int,
                        c_all: int, avg_factor: float):
    """gfl_head_increment_erd.py:225-322.  cls_score [n,C_all] / bbox_pred [n,68] already
    NHWC-flattened over (N, h, w).  Supervised losses touch only the NEW channels [c_old:]."""
    cls_new = cls_score[:, c_old:]
    bg = c_all - c_old
    labels = labels.clone()
    labels[labels == c_all] = bg
    pos = ((labels >= 0) & (labels < bg)).nonzero().squeeze(1)
    score = label_weights.new_zeros(labels.shape)
    if len(pos) > 0:
        pbt = bbox_targets[pos]
        pbp = bbox_pred[pos]
        pc = anchor_centers(anchors[pos]) / stride
        wt = cls_new.detach().sigmoid().max(dim=1)[0][pos]
        corners = integral(pbp)
        dec = distance2bbox(pc, corners)
        tgt = pbt / stride
        score[pos] = bbox_overlaps(dec.detach(), tgt, is_aligned=True)
        target_corners = bbox2distance(pc, tgt, REG_MAX).reshape(-1)
        loss_bbox = giou_loss_module(dec, tgt, wt, 1.0)
        loss_dfl = 0.25 * weight_reduce(distribution_focal_loss(pbp.reshape(-1, REG_MAX + 1), target_corners),
                                        wt[:, None].expand(-1, 4).reshape(-1), 4.0)
        wsum = wt.sum()
    else:
        loss_bbox = bbox_pred.sum() * 0
        loss_dfl = bbox_pred.sum() * 0
        wsum = bbox_pred.new_tensor(0).sum()
    loss_cls = 1.0 * weight_reduce(quality_focal_loss(cls_new, labels, score), label_weights, avg_factor)
    return loss_cls, loss_bbox, loss_dfl, wsum


def distill_loss_single(anchors: Tensor, s_cls_old: Tensor, s_bbox: Tensor, idx_c: Tensor, idx_b: Tensor,
                        t_cls: Tensor, t_bbox: Tensor, w: float, T: float = 10.0, ld_weight: float = 0.25):
    """gfl_head_increment_erd.py:142-223 for ONE image ([A,*] tensors).
    D8: NMS boxes = pixel-unit centres + stride-unit distances (no scaling) -- reproduced."""
    l2 = (s_cls_old[idx_c] - t_cls[idx_c]).pow(2).float().mean()
    loss_dist_cls = w * l2
    dec = distance2bbox(anchor_centers(anchors), integral(t_bbox))
    conf, ids = t_cls.sigmoid().max(dim=-1)
    keep = nms_class_offset(dec[idx_b], conf[idx_b], ids[idx_b], 0.005)
    s_sel = s_bbox[idx_b][keep].reshape(-1, REG_MAX + 1)
    t_sel = t_bbox[idx_b][keep].reshape(-1, REG_MAX + 1)
    wt = s_cls_old[idx_b].detach().sigmoid().max(dim=1)[0][keep]
    kd = kd_kl_div(s_sel, t_sel, T)
    loss_dist_bbox = w * ld_weight * weight_reduce(kd, wt[:, None].expand(-1, 4).reshape(-1), 4.0)
    return loss_dist_cls, loss_dist_bbox, keep


def erd_head_loss(t_cls_maps, t_bbox_maps, s_cls_maps, s_bbox_maps, gt_bboxes: Sequence[Tensor],
                  gt_labels: Sequence[Tensor], metas: Sequence[dict], c_old: int, c_all: int,
                  dist_loss_weight: float = 1.0, world_size: int = 1, return_aux: bool = False,
                  rank_mean_factors: Optional[Tuple[float, float]] = None):
    """GFLIncrementERD.sel_pos + GFLHeadIncrementERD.loss_by_feat
    (gfl_increment_erd.py:165-200; gfl_head_increment_erd.py:334-454).  reduce_mean is the
    identity for world_size 1 (dist_utils.py:59-65).  Data parallel (world > 1): the two normalisers are means over the
    ranks of each rank's LOCAL value (`reduce_mean`, gfl_head_increment_erd.py:390-391 and :406-407, the second one
    clamped to >= 1 AFTER the mean); a multi-rank evaluation runs every rank's batch once for its local values
    (aux['avg_factor'], aux['weight_sum']), averages them and passes `rank_mean_factors = (mean num_pos, mean weight sum)`."""
    N = s_cls_maps[0].shape[0]
    sizes = [tuple(m.shape[-2:]) for m in s_cls_maps]
    nl = [h * w for h, w in sizes]
    t_cls = flatten_levels(t_cls_maps)
    t_bbox = flatten_levels(t_bbox_maps)
    ers = [ers_select_single(t_cls[i], t_bbox[i]) for i in range(N)]
    anchors = torch.cat(grid_anchors(sizes), 0)
    labs, lws, bts, npos = [], [], [], 0
    for i in range(N):
        flags = torch.cat(valid_flags(sizes, metas[i]["pad_shape"]), 0)
        lab, lw, bt, p = get_targets_single(anchors, flags, nl, gt_bboxes[i], gt_labels[i], c_all)
        labs.append(lab); lws.append(lw); bts.append(bt)
        npos += max(p, 1)                    # sampling_result.py:96-100 avg_factor = max(#pos,1)
    labs, lws, bts = torch.stack(labs), torch.stack(lws), torch.stack(bts)
    avg = float(npos) if rank_mean_factors is None else float(rank_mean_factors[0])      # :390-391
    s_cls = flatten_levels(s_cls_maps)
    s_bbox = flatten_levels(s_bbox_maps)
    lc, lb, ld, ws = [], [], [], []
    off = 0
    for l, n_l in enumerate(nl):
        sl = slice(off, off + n_l)
        a = anchors[sl][None].expand(N, -1, -1).reshape(-1, 4)
        r = loss_by_feat_single(a, s_cls[:, sl].reshape(-1, c_all), s_bbox[:, sl].reshape(-1, 68),
                                labs[:, sl].reshape(-1), lws[:, sl].reshape(-1), bts[:, sl].reshape(-1, 4),
                                STRIDES[l], c_old, c_all, avg)
        lc.append(r[0]); lb.append(r[1]); ld.append(r[2]); ws.append(r[3])
        off += n_l
    wsum = float(sum(ws).detach())
    avg2 = max(wsum if rank_mean_factors is None else float(rank_mean_factors[1]), 1.0)  # :406-407 (float(): the reference's .item())
    lb = [x / avg2 for x in lb]
    ld = [x / avg2 for x in ld]
    dc, db, keeps = [], [], []
    for i in range(N):
        a, b, k = distill_loss_single(anchors, s_cls[i, :, :c_old], s_bbox[i], ers[i][0], ers[i][1],
                                      t_cls[i, :, :c_old], t_bbox[i], dist_loss_weight)
        dc.append(a); db.append(b); keeps.append(k)
    losses = dict(loss_cls=lc, loss_bbox=lb, loss_dfl=ld, loss_dist_cls=dc, loss_dist_bbox=db)
    if return_aux:
        aux = dict(ers_cls=[e[0] for e in ers], ers_bbox=[e[1] for e in ers],
                   thr_cls=[e[2] for e in ers], thr_bbox=[e[3] for e in ers], nms_keep=keeps,
                   labels=labs, label_weights=lws, bbox_targets=bts, avg_factor=avg, avg_factor2=avg2,
                   num_pos_local=float(npos), weight_sum_local=wsum)
        return losses, aux
    return losses


def gfl_head_loss(s_cls_maps, s_bbox_maps, gt_bboxes, gt_labels, metas, num_classes: int):
    """Plain GFLHead.loss_by_feat (gfl_head.py:245-406) == ERD supervised part with c_old=0
    (BASELINE.json configs[0]: first_40_cats forward+loss)."""
    N = s_cls_maps[0].shape[0]
    sizes = [tuple(m.shape[-2:]) for m in s_cls_maps]
    nl = [h * w for h, w in sizes]
    anchors = torch.cat(grid_anchors(sizes), 0)
    labs, lws, bts, npos = [], [], [], 0
    for i in range(N):
        flags = torch.cat(valid_flags(sizes, metas[i]["pad_shape"]), 0)
        lab, lw, bt, p = get_targets_single(anchors, flags, nl, gt_bboxes[i], gt_labels[i], num_classes)
        labs.append(lab); lws.append(lw); bts.append(bt)
        npos += max(p, 1)
    labs, lws, bts = torch.stack(labs), torch.stack(lws), torch.stack(bts)
    s_cls = flatten_levels(s_cls_maps)
    s_bbox = flatten_levels(s_bbox_maps)
    lc, lb, ld, ws = [], [], [], []
    off = 0
    for l, n_l in enumerate(nl):
        sl = slice(off, off + n_l)
        a = anchors[sl][None].expand(N, -1, -1).reshape(-1, 4)
        r = loss_by_feat_single(a, s_cls[:, sl].reshape(-1, num_classes), s_bbox[:, sl].reshape(-1, 68),
                                labs[:, sl].reshape(-1), lws[:, sl].reshape(-1), bts[:, sl].reshape(-1, 4),
                                STRIDES[l], 0, num_classes, float(npos))
        lc.append(r[0]); lb.append(r[1]); ld.append(r[2]); ws.append(r[3])
        off += n_l
    avg2 = float(sum(ws).detach().clamp(min=1))
    return dict(loss_cls=lc, loss_bbox=[x / avg2 for x in lb], loss_dfl=[x / avg2 for x in ld])


# ----------------------------------------------------------------------------
# inference: gfl_head.py:408-502, models/utils/misc.py:308-354,
# base_dense_head.py:201-289,424-486, single_stage.py:78-112
# ----------------------------------------------------------------------------
def filter_scores_and_topk(scores: Tensor, score_thr: float, topk: int):
    """misc.py:308-354.  The reference sorts with torch's default (unstable) sort, so tied scores may
    come out in any order there; the restatement (and the HIP path) break ties by ascending
    (anchor, class) index, which is one of the reference's valid outcomes."""
    valid = scores > score_thr
    s = scores[valid]
    vidx = torch.nonzero(valid)
    k = min(topk, vidx.size(0))
    s, order = s.sort(descending=True, stable=True)
    top = vidx[order[:k]]
    return s[:k], top[:, 1], top[:, 0]


def predict_single(cls_maps: Sequence[Tensor], bbox_maps: Sequence[Tensor], img_shape, scale_factor=None,
                   score_thr: float = 0.05, nms_pre: int = 1000, min_bbox_size: float = 0,
                   iou_thr: float = 0.6, max_per_img: int = 100, strides: Sequence[int] = STRIDES,
                   with_nms: bool = True):
    """GFLHead._predict_by_feat_single + BaseDenseHead._bbox_post_process for ONE image.
    cls_maps[l] [C,h,w], bbox_maps[l] [68,h,w].  scale_factor = (w_scale, h_scale) or None (rescale=False).
    Returns (bboxes [D,4], scores [D], labels [D])."""
    sizes = [tuple(m.shape[-2:]) for m in cls_maps]
    anchors = grid_anchors(sizes, strides)
    bb, ss, ll = [], [], []
    for l, (cm, bm) in enumerate(zip(cls_maps, bbox_maps)):
        C = cm.shape[0]
        dist = integral(bm.permute(1, 2, 0)) * strides[l]
        scores = cm.permute(1, 2, 0).reshape(-1, C).sigmoid()
        s, labels, keep = filter_scores_and_topk(scores, score_thr, nms_pre)
        boxes = distance2bbox(anchor_centers(anchors[l])[keep], dist[keep])
        boxes[:, 0::2].clamp_(min=0, max=img_shape[1])      # transforms.py:175-179 (max_shape = img_shape)
        boxes[:, 1::2].clamp_(min=0, max=img_shape[0])
        bb.append(boxes); ss.append(s); ll.append(labels)
    boxes, scores, labels = torch.cat(bb), torch.cat(ss), torch.cat(ll)
    if scale_factor is not None:                              # base_dense_head.py:458-461, transforms.py:411-414
        boxes = boxes * boxes.new_tensor([1 / f for f in scale_factor]).repeat((1, 2))
    if min_bbox_size >= 0:                                    # :470-474
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        ok = (w > min_bbox_size) & (h > min_bbox_size)
        if not ok.all():
            boxes, scores, labels = boxes[ok], scores[ok], labels[ok]
    if with_nms and boxes.numel() > 0:                        # :477-484
        keep = nms_class_offset(boxes, scores, labels, iou_thr)[:max_per_img]
        boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
    return boxes, scores, labels


def predict_by_feat(cls_maps: Sequence[Tensor], bbox_maps: Sequence[Tensor], metas, rescale: bool = False, **cfg):
    """base_dense_head.py:201-289: per image over batched [N,C,h,w] maps."""
    out = []
    for i, m in enumerate(metas):
        out.append(predict_single([c[i] for c in cls_maps], [b[i] for b in bbox_maps], m["img_shape"],
                                  m["scale_factor"] if rescale else None, **cfg))
    return out


def gfl_predict(sd: Dict[str, Tensor], x: Tensor, metas, depth: int = 50, rescale: bool = True, **cfg):
    """SingleStageDetector.predict (single_stage.py:78-112): forward + predict_by_feat."""
    with torch.no_grad():
        cls, bbox = gfl_forward(sd, x, depth)
        return predict_by_feat(cls, bbox, metas, rescale=rescale, **cfg)


def parse_losses(losses: Dict[str, object]) -> Tensor:
    """mmengine BaseModel.parse_losses (external, UNPINNED; D9): tensor -> mean, list -> sum of
    means; total = sum over keys containing 'loss'."""
    total = 0
    for k, v in losses.items():
        if "loss" not in k:
            continue
        if isinstance(v, (list, tuple)):
            total = total + sum(x.mean() for x in v)
        else:
            total = total + v.mean()
    return total


def erd_step_loss(teacher_sd, student_sd, x: Tensor, gt_bboxes, gt_labels, metas, c_old: int, c_all: int,
                  depth: int = 50, dist_loss_weight: float = 1.0, return_aux: bool = False,
                  rank_mean_factors: Optional[Tuple[float, float]] = None):
    """GFLIncrementERD.loss (gfl_increment_erd.py:202-220): teacher fwd, ERS, student fwd, losses."""
    with torch.no_grad():          # D6: numerically identical, teacher params are frozen
        t_cls, t_bbox = gfl_forward(teacher_sd, x, depth)
    s_cls, s_bbox = gfl_forward(student_sd, x, depth)
    return erd_head_loss(t_cls, t_bbox, s_cls, s_bbox, gt_bboxes, gt_labels, metas, c_old, c_all,
                         dist_loss_weight, return_aux=return_aux, rank_mean_factors=rank_mean_factors)


def trainable(name: str) -> bool:
    """frozen_stages=1 (resnet.py:613-629): stem + layer1 frozen; running stats are buffers."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("running_mean", "running_var", "num_batches_tracked", "project"):
        return False
    if name.startswith("backbone.conv1") or name.startswith("backbone.bn1") or name.startswith("backbone.layer1."):
        return False
    return True


def sgd_momentum_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], bufs: Dict[str, Tensor],
                      lr: float, momentum: float = 0.9, weight_decay: float = 1e-4) -> None:
    """torch.optim.SGD semantics (config :112-114): g += wd*p; buf = mom*buf + g (first step buf = g);
    p -= lr*buf."""
    for k, p in params.items():
        g = grads[k] + weight_decay * p
        if k not in bufs:
            bufs[k] = g.clone()
        else:
            bufs[k].mul_(momentum).add_(g)
        p.sub_(lr * bufs[k])


# ----------------------------------------------------------------------------
# synthetic inputs (mmdet/testing/_utils.py:66-75,89-202)
# ----------------------------------------------------------------------------
def rand_bboxes(rng: np.random.RandomState, num_boxes: int, w: int, h: int) -> np.ndarray:
    cx, cy, bw, bh = rng.rand(num_boxes, 4).T
    tl_x = ((cx * w) - (w * bw / 2)).clip(0, w)
    tl_y = ((cy * h) - (h * bh / 2)).clip(0, h)
    br_x = ((cx * w) + (w * bw / 2)).clip(0, w)
    br_y = ((cy * h) + (h * bh / 2)).clip(0, h)
    return np.vstack([tl_x, tl_y, br_x, br_y]).T


def synthetic_batch(n: int, h: int = 800, w: int = 1333, num_new_classes: int = 40, seed: int = 0):
    """uint8 images + 1..9 random boxes + labels in [0, C_new) per image, RandomState(seed)."""
    rng = np.random.RandomState(seed)
    images, boxes, labels = [], [], []
    for _ in range(n):
        images.append(torch.from_numpy(rng.randint(0, 255, size=(3, h, w), dtype=np.uint8)))
        nb = rng.randint(1, 10)
        boxes.append(torch.from_numpy(rand_bboxes(rng, nb, w, h).astype(np.float32)))
        labels.append(torch.from_numpy(rng.randint(0, num_new_classes, size=nb).astype(np.int64)))
    return images, boxes, labels
