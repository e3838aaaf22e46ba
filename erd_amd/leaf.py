"""Stand-alone leaf operators behind the registered loss / coder / assigner modules (``erd_amd/csrc/leaf_ops.hip``).

The training step runs the fused loss kernels; a reference-side caller that invokes the modules one at a time --
``self.loss_cls(pred, (labels, score), weight=..., avg_factor=...)``, ``self.loss_dfl(...)``, ``self.bbox_coder.decode(...)``,
``self.assigner.assign(...)``, ``self.integral(x)`` -- lands here.  Each function mirrors the reference signature and
`weight_reduce_loss` rule (mmdet/models/losses/utils.py:30-65) and is differentiable through a hand-written backward.
No arithmetic happens in torch ops: tensors are only allocated and handed to the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import kernels as K
from ._lib import call

Tensor = torch.Tensor
_EPS32 = float(torch.finfo(torch.float32).eps)
_p, _stream = K._p, K._stream


def _f32c(t: Tensor) -> Tensor:
    K._require_gpu(t)
    if t.dtype != torch.float32:
        raise TypeError(f"fp32 tensors only (got {t.dtype})")
    return t.contiguous()


def _reduction_scale(n_rows: int, reduction: str, avg_factor) -> Optional[float]:
    """the scalar the weighted row sum is multiplied by (None: reduction='none'); weight_reduce_loss, utils.py:30-65"""
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(f"reduction={reduction!r}")
    if avg_factor is None:
        if reduction == "none":
            return None
        return 1.0 / max(n_rows, 1) if reduction == "mean" else 1.0          # (mean of an empty tensor is NaN in torch;
    if reduction == "mean":                                                   #  the callers never reduce empty rows)
        return 1.0 / (float(avg_factor) + _EPS32)
    if reduction == "none":
        return None
    raise ValueError('avg_factor can not be used with reduction="sum"')


def _rows_mul(rows: Tensor, weight: Optional[Tensor], scale: float) -> Tensor:
    out = torch.empty_like(rows)
    call("erd_rows_mul", _p(rows), _p(weight), rows.numel(), float(scale), _p(out), _stream())
    return out


class _RowLoss(torch.autograd.Function):
    """loss = reduce(loss_weight * rows(pred, *fixed) * weight): forward / backward through two C-ABI calls each.
    `spec` = (rows_fn, bwd_fn): rows_fn(pred, fixed) -> rows [n]; bwd_fn(pred, fixed, coef [n]) -> dpred."""

    @staticmethod
    def forward(ctx, pred, weight, scale, loss_weight, spec, *fixed):
        rows = spec[0](pred, fixed)
        ctx.spec, ctx.fixed, ctx.scale, ctx.lw = spec, fixed, scale, loss_weight
        ctx.save_for_backward(pred, weight if weight is not None else pred.new_empty(0))
        ctx.has_w = weight is not None
        if scale is None:          # reduction='none'
            return _rows_mul(rows, weight, loss_weight)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        call("erd_weighted_sum", _p(rows), _p(weight), rows.numel(), C.c_double(scale * loss_weight), _p(out), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        pred, w = ctx.saved_tensors
        w = w if ctx.has_w else None
        n = pred.shape[0]
        g = g.contiguous()
        if ctx.scale is None:
            coef = _rows_mul(g, w, ctx.lw)
        else:
            coef = torch.empty(n, dtype=torch.float32, device=pred.device)
            call("erd_loss_coef", _p(g), _p(w), n, float(ctx.scale * ctx.lw), _p(coef), _stream())
        return (ctx.spec[1](pred, ctx.fixed, coef), None, None, None, None) + (None,) * len(ctx.fixed)


def _row_loss(pred: Tensor, weight: Optional[Tensor], reduction: str, avg_factor, loss_weight: float, spec, *fixed) -> Tensor:
    pred = _f32c(pred)
    n = pred.shape[0]
    if weight is not None:
        weight = _f32c(weight)
        if weight.numel() != n:
            raise ValueError(f"weight has {weight.numel()} entries for {n} rows")
    return _RowLoss.apply(pred, weight, _reduction_scale(n, reduction, avg_factor), float(loss_weight), spec, *fixed)


# ---- QFL (gfocal_loss.py:12-53, 168-249) -----------------------------------------------------------------------------
def _qfl_rows(pred, fixed):
    label, score = fixed
    rows = torch.empty(pred.shape[0], dtype=torch.float32, device=pred.device)
    call("erd_qfl_rows", _p(pred), _p(label), _p(score), pred.shape[0], pred.shape[1], _p(rows), _stream())
    return rows


def _qfl_bwd(pred, fixed, coef):
    label, score = fixed
    d = torch.empty_like(pred)
    call("erd_qfl_bwd", _p(pred), _p(label), _p(score), _p(coef), pred.shape[0], pred.shape[1], _p(d), _stream())
    return d


def quality_focal_loss(pred: Tensor, target: Tuple[Tensor, Tensor], weight=None, beta: float = 2.0, reduction="mean",
                       avg_factor=None, loss_weight: float = 1.0) -> Tensor:
    if beta != 2.0:
        raise NotImplementedError("the QFL kernel is built for beta = 2.0")
    if not isinstance(target, (tuple, list)) or len(target) != 2:
        raise NotImplementedError("target must be the (label, score) tuple (the tensor-target form is for activated inputs)")
    label, score = target
    K._require_gpu(label, score)
    return _row_loss(pred, weight, reduction, avg_factor, loss_weight, (_qfl_rows, _qfl_bwd),
                     label.to(torch.int64).contiguous(), _f32c(score))


# ---- DFL (gfocal_loss.py:143-165, 252-295) ------------------------------------------------------------------------------
def _dfl_rows(pred, fixed):
    rows = torch.empty(pred.shape[0], dtype=torch.float32, device=pred.device)
    call("erd_dfl", _p(pred), _p(fixed[0]), None, pred.shape[0], pred.shape[1], _p(rows), None, _stream())
    return rows


def _dfl_bwd(pred, fixed, coef):
    d = torch.empty_like(pred)
    call("erd_dfl", _p(pred), _p(fixed[0]), _p(coef), pred.shape[0], pred.shape[1], None, _p(d), _stream())
    return d


def distribution_focal_loss(pred: Tensor, label: Tensor, weight=None, reduction="mean", avg_factor=None,
                            loss_weight: float = 1.0) -> Tensor:
    return _row_loss(pred, weight, reduction, avg_factor, loss_weight, (_dfl_rows, _dfl_bwd), _f32c(label))


# ---- KD-KL (kd_loss.py:12-95) -------------------------------------------------------------------------------------------
def _kd_rows(pred, fixed):
    soft, T = fixed
    rows = torch.empty(pred.shape[0], dtype=torch.float32, device=pred.device)
    call("erd_kd_kl_rows", _p(pred), _p(soft), None, pred.shape[0], pred.shape[1], float(T), _p(rows), None, _stream())
    return rows


def _kd_bwd(pred, fixed, coef):
    soft, T = fixed
    d = torch.empty_like(pred)
    call("erd_kd_kl_rows", _p(pred), _p(soft), _p(coef), pred.shape[0], pred.shape[1], float(T), None, _p(d), _stream())
    return d


def knowledge_distillation_kl_div_loss(pred: Tensor, soft_label: Tensor, weight=None, reduction="mean", avg_factor=None,
                                       T: float = 10, loss_weight: float = 1.0) -> Tensor:
    if pred.shape != soft_label.shape:
        raise AssertionError("pred and soft_label must share their shape")
    return _row_loss(pred, weight, reduction, avg_factor, loss_weight, (_kd_rows, _kd_bwd), _f32c(soft_label.detach()), float(T))


# ---- GIoU (iou_loss.py:110-126, 463-528) + overlaps ---------------------------------------------------------------------
def _giou_rows(pred, fixed):
    target, eps = fixed
    rows = torch.empty(pred.shape[0], dtype=torch.float32, device=pred.device)
    call("erd_giou", _p(pred), _p(target), None, pred.shape[0], float(eps), _p(rows), None, _stream())
    return rows


def _giou_bwd(pred, fixed, coef):
    target, eps = fixed
    d = torch.empty_like(pred)
    call("erd_giou", _p(pred), _p(target), _p(coef), pred.shape[0], float(eps), None, _p(d), _stream())
    return d


def giou_loss(pred: Tensor, target: Tensor, weight=None, eps: float = 1e-6, reduction="mean", avg_factor=None,
              loss_weight: float = 1.0) -> Tensor:
    if weight is not None and weight.dim() > 1:       # iou_loss.py:511-515: per-coordinate weights -> per-box mean
        if weight.shape != pred.shape:
            raise AssertionError
        weight = weight.mean(-1)
    # (the reference's early-out for an all-zero weight returns (pred * weight).sum() == 0: the weighted sum below gives
    #  the same 0 and the same zero gradient without reading the weights back)
    return _row_loss(pred, weight, reduction, avg_factor, loss_weight, (_giou_rows, _giou_bwd), _f32c(target.detach()), float(eps))


def bbox_overlaps(bboxes1: Tensor, bboxes2: Tensor, mode: str = "iou", is_aligned: bool = False, eps: float = 1e-6) -> Tensor:
    """structures/bbox/bbox_overlaps.py:13-199 for 2-D [m,4] / [n,4] inputs (a trailing score column is ignored)"""
    if mode not in ("iou", "giou"):
        raise NotImplementedError(f"mode {mode!r}: 'iou' and 'giou' are built ('iof' is off the ERD path)")
    b1, b2 = _f32c(bboxes1[..., :4]), _f32c(bboxes2[..., :4])
    if b1.dim() != 2 or b2.dim() != 2:
        raise NotImplementedError("2-D box tensors only")
    m, n = b1.shape[0], b2.shape[0]
    if is_aligned:
        if m != n:
            raise AssertionError
        out = torch.empty(m, dtype=torch.float32, device=b1.device)
    else:
        out = torch.empty((m, n), dtype=torch.float32, device=b1.device)
    call("erd_bbox_overlaps", _p(b1), _p(b2), m, n, int(is_aligned), int(mode == "giou"), float(eps), _p(out), _stream())
    return out


# ---- Integral (gfl_head.py:29-62) ---------------------------------------------------------------------------------------
class _Integral(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nb):
        rows = x.numel() // nb
        y = torch.empty(rows, dtype=torch.float32, device=x.device)
        call("erd_integral", _p(x), None, rows, nb, _p(y), None, _stream())
        ctx.save_for_backward(x)
        ctx.nb = nb
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("erd_integral", _p(x), _p(dy.contiguous()), x.numel() // ctx.nb, ctx.nb, None, _p(dx), _stream())
        return dx, None


def integral(x: Tensor, reg_max: int = 16) -> Tensor:
    """[n, 4 (reg_max + 1)] side distributions -> [n, 4] expected distances"""
    x = _f32c(x)
    return _Integral.apply(x, reg_max + 1).reshape(-1, 4)


# ---- DistancePointBBoxCoder (distance_point_bbox_coder.py:28-85, transforms.py:147-230) --------------------------------
class _Distance2BBox(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, dist, max_h, max_w):
        out = torch.empty_like(dist)
        call("erd_distance2bbox", _p(points), _p(dist), None, dist.shape[0], float(max_h), float(max_w), _p(out), None, _stream())
        ctx.save_for_backward(points, dist)
        ctx.lim = (max_h, max_w)
        return out

    @staticmethod
    def backward(ctx, g):
        points, dist = ctx.saved_tensors
        dd = torch.empty_like(dist)
        call("erd_distance2bbox", _p(points), _p(dist), _p(g.contiguous()), dist.shape[0], float(ctx.lim[0]), float(ctx.lim[1]),
             None, _p(dd), _stream())
        return None, dd, None, None


def distance2bbox(points: Tensor, distance: Tensor, max_shape: Optional[Sequence[int]] = None) -> Tensor:
    points, distance = _f32c(points[..., :2]), _f32c(distance)
    if points.dim() != 2 or distance.shape != (points.shape[0], 4):
        raise NotImplementedError("[n,2] points and [n,4] distances")
    mh, mw = (-1.0, -1.0) if max_shape is None else (float(max_shape[0]), float(max_shape[1]))
    return _Distance2BBox.apply(points, distance, mh, mw)


def bbox2distance(points: Tensor, bbox: Tensor, max_dis: Optional[float] = None, eps: float = 0.1) -> Tensor:
    points, bbox = _f32c(points[..., :2]), _f32c(bbox)
    out = torch.empty_like(bbox)
    call("erd_bbox2distance", _p(points), _p(bbox), bbox.shape[0], -1.0 if max_dis is None else float(max_dis), float(eps), _p(out),
         _stream())
    return out
