"""Host-side mirror of the reference's operator/plugin interface for the ERD path: the same registered
class names, constructor arguments, state-dict keys and ``forward(inputs, data_samples, mode)`` contract
(SURVEY.md 8(b)), with every tensor op dispatched to the HIP library.  There is no CPU execution path:
calling a module with CPU tensors raises.

Reference classes mirrored (all under /root/reference/mmdet/models):
  ResNet/Bottleneck      backbones/resnet.py:97-302,305-657 ; layers/res_layer.py:12-109
  FPN                    necks/fpn.py:15-221
  GFLHead                dense_heads/gfl_head.py:65-230 (+ anchor_head.py, base_dense_head.py train parts)
  GFLHeadIncrementERD    dense_heads/gfl_head_increment_erd.py:57-484
  GFL / GFLIncrementERD  detectors/{base,single_stage,gfl,gfl_increment_erd}.py
  losses / task utils    losses/{gfocal_loss,kd_loss,iou_loss}.py, task_modules/*
"""
from __future__ import annotations

import math
import os
import sys
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .config import Config, ConfigDict
from .registry import MODELS, TASK_UTILS
from .structures import DetDataSample, InstanceData, unpack_gt_instances

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------------
# parameter holders (state-dict ABI only; compute is in erd_amd.functional)
# ---------------------------------------------------------------------------------------------------
def _conv_weight(cout: int, cin: int, k: int) -> nn.Parameter:
    """logical OIHW, physical [O][kh][kw][I] (channels_last) -- what the kernels read in place."""
    w = torch.empty((cout, k, k, cin), dtype=torch.float32).permute(0, 3, 1, 2)
    return nn.Parameter(w)


class ConvHolder(nn.Module):
    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, padding: int = 0, bias: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        self.weight = _conv_weight(cout, cin, k)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self, mode: str = "kaiming", std: float = 0.01, bias: float = 0.0):
        with torch.no_grad():
            if mode == "kaiming":      # resnet.py init_weights: kaiming_init(mode='fan_out', relu)
                fan_out = self.out_channels * self.k * self.k
                self.weight.normal_(0, math.sqrt(2.0 / fan_out))
            elif mode == "xavier":     # fpn.py init_cfg Xavier uniform
                fan_in, fan_out = self.in_channels * self.k ** 2, self.out_channels * self.k ** 2
                a = math.sqrt(6.0 / (fan_in + fan_out))
                self.weight.uniform_(-a, a)
            else:                      # head: Normal(std)
                self.weight.normal_(0, std)
            if self.bias is not None:
                self.bias.fill_(bias)


class FrozenStatBN(nn.Module):
    """BatchNorm2d parameters/buffers; statistics are never updated (norm_eval=True, resnet.py:648-657)."""

    def __init__(self, c: int, eps: float = 1e-5, requires_grad: bool = True):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c), requires_grad=requires_grad)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=requires_grad)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class GNHolder(nn.Module):
    def __init__(self, c: int, groups: int = 32, eps: float = 1e-5):
        super().__init__()
        self.num_groups, self.eps = groups, eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule key layout: `.conv.{weight,bias}` (+ `.gn.{weight,bias}`)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm_cfg=None, init: str = "xavier", **kw):
        super().__init__()
        self.conv = ConvHolder(cin, cout, k, stride, padding, bias=norm_cfg is None)
        self.conv.reset_parameters(init, kw.get("std", 0.01))
        if norm_cfg is not None:
            assert norm_cfg["type"] == "GN", "only GN is used by the GFL head (gfl_head.py:109-110)"
            self.gn = GNHolder(cout, norm_cfg.get("num_groups", 32))


class Scale(nn.Module):
    def __init__(self, scale: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))


def _nhwc(x: Tensor) -> Tensor:
    """logical NCHW (any memory format) -> NHWC contiguous map; free for channels_last inputs."""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _nchw(m: Tensor) -> Tensor:
    return m.permute(0, 3, 1, 2)


def _gpu_only(x: Tensor, who: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{who}: erd_amd runs on the MI355X HIP path only; got a {x.device} tensor "
                           "(there is no CPU fallback -- the CPU restatement lives in oracle/ and is test-only)")


# ---------------------------------------------------------------------------------------------------
# ResNet
# ---------------------------------------------------------------------------------------------------
# 'torchvision://<name>' in the reference resolves through mmengine's `get_torchvision_models()`, which for every
# torchvision >= 0.13 reads the frozen torchvision-0.12 URL table (mmengine/hub/torchvision_0.12.json): ONE fixed file
# per name.  These are the basenames of those URLs for the ResNets this package builds.
_TORCHVISION_FILES = {"resnet18": "resnet18-f37072fd.pth", "resnet34": "resnet34-b627a593.pth",
                      "resnet50": "resnet50-0676ba61.pth", "resnet101": "resnet101-63fe2227.pth",
                      "resnet152": "resnet152-394f9c45.pth"}


def _resolve_pretrained(checkpoint: str) -> Optional[str]:
    """local path of an init_cfg checkpoint, or None.  'torchvision://resnet50' -> the exact file the reference's
    scheme downloads (resnet50-0676ba61.pth) under $ERD_PRETRAINED_DIR or <torch hub dir>/checkpoints.  Only when that
    file is absent does a `<name>*.pth` glob apply, and then with a warning that names the file that was picked (a
    directory may also hold IMAGENET1K_V2 weights or a user's resnet50_custom.pth)."""
    if os.path.isfile(checkpoint):
        return checkpoint
    if "://" not in checkpoint:
        return None
    name = checkpoint.split("://", 1)[1]
    import glob
    dirs = [d for d in (os.environ.get("ERD_PRETRAINED_DIR"), os.path.join(torch.hub.get_dir(), "checkpoints"))
            if d and os.path.isdir(d)]
    exact = _TORCHVISION_FILES.get(name) if checkpoint.startswith("torchvision://") else None
    if exact:
        for d in dirs:
            if os.path.isfile(os.path.join(d, exact)):
                return os.path.join(d, exact)
    for d in dirs:
        hits = sorted(glob.glob(os.path.join(d, name + "*.pth")))
        if hits:
            import warnings
            warnings.warn(f"{checkpoint!r}: {exact or 'the canonical file'} not found under {dirs}; using {hits[0]} "
                          f"(first of {len(hits)} files matching {name}*.pth)", RuntimeWarning)
            return hits[0]
    return None


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool, bn_requires_grad: bool = True):
        super().__init__()
        self.conv1 = ConvHolder(inplanes, planes, 1)
        self.bn1 = FrozenStatBN(planes, requires_grad=bn_requires_grad)
        self.conv2 = ConvHolder(planes, planes, 3, stride, 1)     # style='pytorch': stride on the 3x3
        self.bn2 = FrozenStatBN(planes, requires_grad=bn_requires_grad)
        self.conv3 = ConvHolder(planes, planes * 4, 1)
        self.bn3 = FrozenStatBN(planes * 4, requires_grad=bn_requires_grad)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(ConvHolder(inplanes, planes * 4, 1, stride),
                                            FrozenStatBN(planes * 4, requires_grad=bn_requires_grad))

    @staticmethod
    def _cba(x, conv: ConvHolder, bn: FrozenStatBN, res, relu: bool):
        return Fn.ConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, res,
                                  conv.k, conv.stride, conv.padding, relu, bn.eps)

    def _params(self):
        mods = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        if self.downsample is not None:
            mods.append((self.downsample[0], self.downsample[1]))
        out = []
        for c, b in mods:
            out += [c.weight, b.weight, b.bias, b.running_mean, b.running_var]
        return out

    def forward(self, x: Tensor) -> Tensor:      # NHWC map
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return Fn.BottleneckFn.apply(x, self.stride, self.bn1.eps, *self._params())
        out = self._cba(x, self.conv1, self.bn1, None, True)
        out = self._cba(out, self.conv2, self.bn2, None, True)
        identity = x
        if self.downsample is not None:
            identity = self._cba(x, self.downsample[0], self.downsample[1], None, False)
        return self._cba(out, self.conv3, self.bn3, identity, True)


@MODELS.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth: int, in_channels: int = 3, stem_channels: Optional[int] = None, base_channels: int = 64,
                 num_stages: int = 4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3),
                 style: str = "pytorch", deep_stem: bool = False, avg_down: bool = False, frozen_stages: int = -1,
                 conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), norm_eval: bool = True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp: bool = False,
                 zero_init_residual: bool = True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f"invalid depth {depth} for resnet (bottleneck depths 50/101/152 are built)")
        unsupported = dict(in_channels=(in_channels, 3), base_channels=(base_channels, 64), style=(style, "pytorch"),
                           deep_stem=(deep_stem, False), avg_down=(avg_down, False), dcn=(dcn, None),
                           plugins=(plugins, None), conv_cfg=(conv_cfg, None), with_cp=(with_cp, False))
        for k, (v, want) in unsupported.items():
            if v != want:
                raise NotImplementedError(f"ResNet({k}={v!r}) is outside the ERD hot path (only {want!r})")
        assert norm_cfg["type"] == "BN" and tuple(strides[:num_stages]) == (1, 2, 2, 2)[:num_stages]
        if not norm_eval:
            raise NotImplementedError("norm_eval=False (batch statistics) is not on the ERD path "
                                      "(configs/gfl_increment/*.py:43 norm_eval=True)")
        self.depth, self.num_stages, self.out_indices = depth, num_stages, tuple(out_indices)
        self.frozen_stages, self.norm_eval, self.init_cfg = frozen_stages, norm_eval, init_cfg
        bn_rg = norm_cfg.get("requires_grad", True)
        self.conv1 = ConvHolder(3, 64, 7, 2, 3)
        self.bn1 = FrozenStatBN(64, requires_grad=bn_rg)
        inplanes = 64
        self.res_layers = []
        for i, nblk in enumerate(self.arch_settings[depth][:num_stages]):
            planes = 64 * 2 ** i
            blocks = []
            for b in range(nblk):
                blocks.append(Bottleneck(inplanes, planes, strides[i] if b == 0 else 1, b == 0, bn_rg))
                inplanes = planes * 4
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()

    def _freeze_stages(self):
        """resnet.py:613-629."""
        if self.frozen_stages >= 0:
            for p in list(self.conv1.parameters()) + list(self.bn1.parameters()):
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, f"layer{i}").parameters():
                p.requires_grad = False

    def train(self, mode: bool = True):
        super().train(mode)
        self._freeze_stages()
        return self

    def init_weights(self):
        """mmengine BaseModule.init_weights with `init_cfg=dict(type='Pretrained', checkpoint=...)` (the shipped
        configs: 'torchvision://resnet50', configs/gfl_increment/*first_40_cats.py:45): load the ImageNet backbone.
        torchvision's keys map one to one onto conv1 / bn1 / layerX.Y.{convZ,bnZ,downsample.{0,1}}; `fc.*` is dropped.
        A 'torchvision://name' checkpoint is looked up on disk only ($ERD_PRETRAINED_DIR, then torch hub's checkpoint
        directory) -- there is no download.  An unresolvable checkpoint RAISES: training a base detector on a frozen
        random stem + layer1 with mean-0 / var-1 BN statistics cannot reproduce the reference's model
        (set ERD_ALLOW_RANDOM_BACKBONE=1 to continue on the constructors' kaiming weights, with a warning)."""
        cfg = self.init_cfg
        if not cfg:
            return      # ConvHolder / FrozenStatBN constructors already hold the reference's default init
        if cfg.get("type") != "Pretrained":
            raise NotImplementedError(f"ResNet.init_cfg type {cfg.get('type')!r}: only 'Pretrained' is built")
        path = _resolve_pretrained(str(cfg["checkpoint"]))
        if path is None:
            msg = (f"ResNet init_cfg: cannot resolve {cfg['checkpoint']!r} on disk (no network); put the file under "
                   f"$ERD_PRETRAINED_DIR or pass backbone.init_cfg.checkpoint=/path/to/resnet.pth")
            if os.environ.get("ERD_ALLOW_RANDOM_BACKBONE", "0") != "1":
                raise FileNotFoundError(msg)
            import warnings
            warnings.warn(msg + " -- continuing with RANDOM (kaiming) backbone weights", RuntimeWarning)
            return
        sd = torch.load(path, map_location="cpu", weights_only=True)      # a plain state dict: no pickled code is run
        sd = sd.get("state_dict", sd)
        prefix = cfg.get("prefix")
        if prefix:
            sd = {k[len(prefix):].lstrip("."): v for k, v in sd.items() if k.startswith(prefix)}
        sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
        own = self.state_dict()
        missing = [k for k in own if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in own]
        if missing or unexpected:
            raise RuntimeError(f"{path}: not a ResNet-{self.depth} state dict (missing {missing[:4]}..., "
                               f"unexpected {unexpected[:4]}...)")
        self.load_state_dict({k: v for k, v in sd.items()}, strict=False)
        self._freeze_stages()

    def trunk(self, x: Tensor):
        """the frozen front of the network (resnet.py:613-629 `_freeze_stages`): stem + max-pool + the first
        `frozen_stages` stages, run without autograd.  Returns (h, outs): the NHWC map the first trainable stage reads
        and the outputs of the frozen stages listed in `out_indices`.  A detector whose student and teacher hold the same
        frozen trunk computes it once and feeds both (GFLIncrementERD.shares_trunk)."""
        _gpu_only(x, "ResNet.forward")
        if x.dtype != torch.float32:
            raise TypeError("fp32 inputs only")
        x = x.contiguous()
        outs = []
        with torch.no_grad():
            scale, shift = Fn._bn_fold_cached(self.bn1.weight, self.bn1.bias, self.bn1.running_mean,
                                              self.bn1.running_var, self.bn1.eps)
            h = K.stem(x, Fn.ohwi(self.conv1.weight), scale, shift)
            for i in range(min(max(self.frozen_stages, 0), len(self.res_layers))):
                for blk in getattr(self, self.res_layers[i]):
                    h = blk(h)
                if i in self.out_indices:
                    outs.append(_nchw(h))
        return h, outs

    def forward(self, x: Tensor, trunk=None) -> Tuple[Tensor, ...]:
        """x: [N,3,H,W] NCHW fp32 -> tuple of logical-NCHW (channels_last) stage outputs.  `trunk`: the result of
        `trunk(x)` of a network with the SAME frozen front (then x is not read)."""
        h, outs = self.trunk(x) if trunk is None else trunk
        outs = list(outs)
        for i, name in enumerate(self.res_layers):
            if (i + 1) <= self.frozen_stages:
                continue
            layer = getattr(self, name)
            ctxm = torch.no_grad() if not torch.is_grad_enabled() else torch.enable_grad()
            with ctxm:
                if Fn.RES_LAYER_NODE and torch.is_grad_enabled() and all(
                        any(p.requires_grad for p in blk.parameters()) for blk in layer):
                    # the stage as ONE autograd node: its backward hands each block's dz3 straight to the block before it
                    params, counts = [], []
                    for blk in layer:
                        pp = blk._params()
                        params += pp
                        counts.append(len(pp))
                    h = Fn.ResLayerFn.apply(h, tuple(blk.stride for blk in layer), layer[0].bn1.eps, tuple(counts), *params)
                else:
                    for blk in layer:
                        h = blk(h)
            if i in self.out_indices:
                outs.append(_nchw(h))
        return tuple(outs)

    def trunk_tensors(self) -> List[Tensor]:
        """every parameter / buffer the frozen front reads (for the equality check of a shared trunk)"""
        mods = [self.conv1, self.bn1] + [getattr(self, n) for n in self.res_layers[:max(self.frozen_stages, 0)]]
        out = []
        for m in mods:
            out += [t for _, t in sorted(m.state_dict().items()) if t.dtype == torch.float32]
        return out


# ---------------------------------------------------------------------------------------------------
# FPN
# ---------------------------------------------------------------------------------------------------
@MODELS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels: List[int], out_channels: int, num_outs: int, start_level: int = 0,
                 end_level: int = -1, add_extra_convs: Union[bool, str] = False, relu_before_extra_convs: bool = False,
                 no_norm_on_lateral: bool = False, conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=None,
                 init_cfg=None):
        super().__init__()
        if isinstance(add_extra_convs, bool) and add_extra_convs:
            add_extra_convs = "on_input"
        if (add_extra_convs != "on_output" or relu_before_extra_convs or norm_cfg is not None or act_cfg is not None
                or end_level not in (-1, len(in_channels) - 1)
                or (upsample_cfg not in (None, dict(mode="nearest")))):
            raise NotImplementedError("FPN variant outside the ERD hot path (built: add_extra_convs='on_output', "
                                      "no norm/act, nearest top-down: configs/gfl_increment/*.py:47-53)")
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.start_level = start_level
        self.backbone_end_level = len(in_channels)
        nlat = self.backbone_end_level - start_level
        if nlat != 3 or num_outs != 5 or out_channels != 256:
            raise NotImplementedError("the fused FPN output kernel sequence is built for 3 laterals + P6/P7, 256 ch")
        self.lateral_convs = nn.ModuleList(
            [ConvModule(in_channels[i], out_channels, 1) for i in range(start_level, self.backbone_end_level)])
        fpn = [ConvModule(out_channels, out_channels, 3, padding=1) for _ in range(nlat)]
        fpn += [ConvModule(out_channels, out_channels, 3, stride=2, padding=1) for _ in range(num_outs - nlat)]
        self.fpn_convs = nn.ModuleList(fpn)

    def forward_cat(self, inputs: Sequence[Tensor]) -> Tuple[Tensor, List[Tuple[int, int]]]:
        """-> ([N,A,256] level-concatenated NHWC buffer, [(h,w)]*5)"""
        assert len(inputs) == len(self.in_channels)
        lats = []
        for i, lc in enumerate(self.lateral_convs):
            x = _nhwc(inputs[i + self.start_level])
            _gpu_only(x, "FPN.forward")
            lats.append(Fn.ConvBias.apply(x, lc.conv.weight, lc.conv.bias, 1, 1, 0))
        for i in range(len(lats) - 1, 0, -1):
            lats[i - 1] = Fn.UpsampleAdd.apply(lats[i - 1], lats[i])
        ws = [m.conv.weight for m in self.fpn_convs]
        bs = [m.conv.bias for m in self.fpn_convs]
        cat = Fn.FPNOutputs.apply(lats[0], lats[1], lats[2], *ws, *bs)
        sizes = [(l.shape[1], l.shape[2]) for l in lats]
        h, w = sizes[-1]
        for _ in range(2):
            h, w = K.conv_out_size(h, 3, 2, 1), K.conv_out_size(w, 3, 2, 1)
            sizes.append((h, w))
        return cat, sizes

    def forward(self, inputs: Sequence[Tensor]) -> Tuple[Tensor, ...]:
        cat, sizes = self.forward_cat(inputs)
        return tuple(_nchw(v) for v in K.level_views(cat, sizes))


# ---------------------------------------------------------------------------------------------------
# task utils (registered so that every `type=` of the configs resolves)
# ---------------------------------------------------------------------------------------------------
@TASK_UTILS.register_module()
class AnchorGenerator:
    """anchor_generator.py:15-476 restricted to what GFL uses: one square base anchor per location
    (ratios=[1.0], scales_per_octave=1, center_offset=0)."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.0, use_box_type=False):
        if list(ratios) != [1.0] or scales_per_octave != 1 or octave_base_scale is None or scales is not None \
                or center_offset != 0.0 or centers is not None or base_sizes is not None:
            raise NotImplementedError("AnchorGenerator variant outside the ERD hot path "
                                      "(built: ratios=[1.0], octave_base_scale=k, scales_per_octave=1)")
        self.strides = [int(s if not isinstance(s, (tuple, list)) else s[0]) for s in strides]
        self.octave_base_scale = int(octave_base_scale)
        self._cache: Dict[tuple, Tensor] = {}

    @property
    def num_levels(self) -> int:
        return len(self.strides)

    @property
    def num_base_priors(self) -> List[int]:
        return [1] * len(self.strides)

    def grid_priors_cat(self, featmap_sizes, device) -> Tensor:
        key = (tuple(map(tuple, featmap_sizes)), str(device))
        if key not in self._cache:
            self._cache[key] = K.grid_anchors([tuple(s) for s in featmap_sizes], self.strides, device,
                                              self.octave_base_scale)
        return self._cache[key]

    def grid_priors(self, featmap_sizes, dtype=torch.float32, device="cuda") -> List[Tensor]:
        cat = self.grid_priors_cat(featmap_sizes, device)
        return list(torch.split(cat, [h * w for h, w in featmap_sizes], 0))

    def valid_flags(self, featmap_sizes, pad_shape, device="cuda") -> List[Tensor]:
        """anchor_generator.py:415-476 (host integers -> flags; metadata only)."""
        out = []
        for (fh, fw), s in zip(featmap_sizes, self.strides):
            h, w = pad_shape[:2]
            vh, vw = min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw)
            f = np.zeros((fh, fw), dtype=np.uint8)
            f[:vh, :vw] = 1
            out.append(torch.from_numpy(f.reshape(-1)))
        return out


class AssignResult:
    """assigners/assign_result.py:8-50: `gt_inds` 0 = unassigned / g + 1, `max_overlaps`, `labels` (-1 = unassigned)"""

    def __init__(self, num_gts: int, gt_inds: Tensor, max_overlaps: Tensor, labels: Tensor):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self) -> int:
        return len(self.gt_inds)


class SamplingResult:
    """samplers/sampling_result.py:86-116 (the fields the GFL target code reads)"""

    def __init__(self, pos_inds, neg_inds, priors, gt_bboxes, assign_result, gt_flags, avg_factor_with_neg: bool = True):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.num_pos, self.num_neg = max(pos_inds.numel(), 1), max(neg_inds.numel(), 1)
        self.avg_factor_with_neg = avg_factor_with_neg
        self.avg_factor = self.num_pos + self.num_neg if avg_factor_with_neg else self.num_pos
        self.pos_priors, self.neg_priors = priors[pos_inds], priors[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_labels = assign_result.labels[pos_inds]
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds.long()]


@TASK_UTILS.register_module()
class ATSSAssigner:
    def __init__(self, topk: int, alpha=None, iou_calculator=dict(type="BboxOverlaps2D"), ignore_iof_thr: float = -1):
        if alpha is not None or ignore_iof_thr > 0:
            raise NotImplementedError("cost-based / ignore-region ATSS is outside the ERD hot path")
        self.topk = topk
        self.alpha = alpha
        self.iou_calculator = TASK_UTILS.build(iou_calculator) if isinstance(iou_calculator, dict) else iou_calculator

    def assign(self, pred_instances, num_level_priors: Sequence[int], gt_instances, gt_instances_ignore=None) -> AssignResult:
        """assigners/atss_assigner.py:74-254 for one image (the batched targets of the training step come from the same
        kernels through `GFLHead._targets`): per level the `topk` priors closest to each gt centre are candidates, the
        threshold is mean + std of their IoUs, positives have their centre inside the gt, conflicts go to the larger
        IoU.  Returns gt_inds / max_overlaps (-1e8 where unassigned, as the reference leaves it) / labels."""
        priors = pred_instances.priors[:, :4].contiguous()
        gt_bboxes, gt_labels = gt_instances.bboxes, gt_instances.labels
        _gpu_only(priors, "ATSSAssigner.assign")
        A, G = priors.shape[0], gt_bboxes.shape[0]
        if sum(int(v) for v in num_level_priors) != A:
            raise ValueError("num_level_priors does not add up to the number of priors")
        dev = priors.device
        if G == 0 or A == 0:          # atss_assigner.py:120-134
            return AssignResult(G, torch.zeros(A, dtype=torch.int64, device=dev), torch.zeros(A, dtype=torch.float32, device=dev),
                                torch.full((A,), -1, dtype=torch.int64, device=dev))
        gt_off = torch.tensor([0, G], dtype=torch.int32, device=dev)
        glab = gt_labels.to(torch.int64).contiguous()
        K.atss_assign(priors, None, [(int(v), 1) for v in num_level_priors], gt_bboxes.float().contiguous(), glab, gt_off, 1, G,
                      num_classes=1 << 30, topk=self.topk)
        ws = K.workspace("atss", A * 8, dev)         # the (IoU, gt) keys the assignment leaves behind
        gt_inds = torch.empty(A, dtype=torch.int64, device=dev)
        max_ov = torch.empty(A, dtype=torch.float32, device=dev)
        labels = torch.empty(A, dtype=torch.int64, device=dev)
        K.call("erd_atss_result", K._p(ws), K._p(glab), A, K._p(gt_inds), K._p(max_ov), K._p(labels), K._stream())
        return AssignResult(G, gt_inds, max_ov, labels)


@TASK_UTILS.register_module()
class BboxOverlaps2D:
    def __init__(self, scale: float = 1.0, dtype=None):
        if scale != 1.0 or dtype not in (None, "fp32"):
            raise NotImplementedError("BboxOverlaps2D(scale, dtype='fp16') is outside the ERD hot path")

    def __call__(self, bboxes1: Tensor, bboxes2: Tensor, mode: str = "iou", is_aligned: bool = False) -> Tensor:
        """assigners/iou2d_calculator.py:16-62: [m, 4|5] x [n, 4|5] -> [m, n] (or [m] aligned)"""
        from . import leaf
        return leaf.bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)


@TASK_UTILS.register_module()
class DistancePointBBoxCoder:
    def __init__(self, clip_border: bool = True, use_box_type: bool = False):
        if use_box_type:
            raise NotImplementedError("box-type outputs are outside the ERD hot path")
        self.clip_border = clip_border

    def encode(self, points: Tensor, gt_bboxes: Tensor, max_dis: Optional[float] = None, eps: float = 0.1) -> Tensor:
        """coders/distance_point_bbox_coder.py:28-51 -> bbox2distance (transforms.py:201-230)"""
        from . import leaf
        assert points.size(0) == gt_bboxes.size(0) and points.size(-1) == 2 and gt_bboxes.size(-1) == 4
        return leaf.bbox2distance(points, gt_bboxes, max_dis, eps)

    def decode(self, points: Tensor, pred_bboxes: Tensor, max_shape=None) -> Tensor:
        """coders/distance_point_bbox_coder.py:53-85 -> distance2bbox (transforms.py:147-198)"""
        from . import leaf
        assert points.size(0) == pred_bboxes.size(0) and points.size(-1) == 2 and pred_bboxes.size(-1) == 4
        return leaf.distance2bbox(points, pred_bboxes, max_shape if self.clip_border else None)


@TASK_UTILS.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result: AssignResult, pred_instances, gt_instances, *args, **kwargs) -> SamplingResult:
        """samplers/pseudo_sampler.py:26-60: every assigned prior is a positive, every other one a negative (index
        plumbing: `nonzero` reads the counts back, exactly as the reference's own call does)"""
        gt_bboxes, priors = gt_instances.bboxes, pred_instances.priors
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        flags = priors.new_zeros(priors.shape[0], dtype=torch.uint8)
        return SamplingResult(pos, neg, priors, gt_bboxes, assign_result, flags, avg_factor_with_neg=False)


class _LossCfg(nn.Module):
    """The loss modules carry their hyper-parameters for the fused HIP loss kernels of the training step
    (erd_gfl_losses_*, erd_kd_kl*), and a `forward` with the reference's signature for callers that invoke them one at a
    time (erd_amd/leaf.py -> erd_amd/csrc/leaf_ops.hip)."""

    reduction = "mean"

    def _reduction(self, override):
        assert override in (None, "none", "mean", "sum")
        return override if override else self.reduction


@MODELS.register_module()
class QualityFocalLoss(_LossCfg):
    def __init__(self, use_sigmoid=True, beta=2.0, reduction="mean", loss_weight=1.0, activated=False):
        super().__init__()
        if not use_sigmoid or beta != 2.0 or activated:
            raise NotImplementedError("QFL kernel is built for use_sigmoid=True, beta=2.0, activated=False")
        self.beta, self.loss_weight, self.reduction = beta, loss_weight, reduction

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """losses/gfocal_loss.py:205-249: pred [n, C] logits, target = (labels [n] with C = background, scores [n])"""
        from . import leaf
        return leaf.quality_focal_loss(pred, target, weight, self.beta, self._reduction(reduction_override), avg_factor,
                                       self.loss_weight)


@MODELS.register_module()
class DistributionFocalLoss(_LossCfg):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        """losses/gfocal_loss.py:271-295: pred [m, reg_max + 1] logits, target [m] distances in [0, reg_max)"""
        from . import leaf
        return leaf.distribution_focal_loss(pred, target, weight, self._reduction(reduction_override), avg_factor, self.loss_weight)


@MODELS.register_module()
class GIoULoss(_LossCfg):
    def __init__(self, eps=1e-6, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.eps, self.loss_weight, self.reduction = eps, loss_weight, reduction

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        """losses/iou_loss.py:482-528: pred / target [n, 4] xyxy"""
        from . import leaf
        return leaf.giou_loss(pred, target, weight, self.eps, self._reduction(reduction_override), avg_factor, self.loss_weight)


@MODELS.register_module()
class KnowledgeDistillationKLDivLoss(_LossCfg):
    def __init__(self, reduction="mean", loss_weight=1.0, T=10):
        super().__init__()
        assert T >= 1
        self.loss_weight, self.T, self.reduction = loss_weight, T, reduction

    def forward(self, pred, soft_label, weight=None, avg_factor=None, reduction_override=None):
        """losses/kd_loss.py:61-95: pred / soft_label [m, bins] logits"""
        from . import leaf
        return leaf.knowledge_distillation_kl_div_loss(pred, soft_label, weight, self._reduction(reduction_override), avg_factor,
                                                       self.T, self.loss_weight)


@MODELS.register_module()
class CrossEntropyLoss(_LossCfg):
    """built (never called) by GFLHead.__init__ as `loss_cls_for_replay_v3` (gfl_head.py:151)."""

    def __init__(self, **kwargs):
        super().__init__()


class Integral(nn.Module):
    """holds the `integral.project` buffer of the checkpoint ABI (gfl_head.py:29-62).  Inside the training step the
    expectation is computed by the fused loss / NMS kernels; `forward` is the stand-alone operator."""

    def __init__(self, reg_max: int = 16):
        super().__init__()
        self.reg_max = reg_max
        self.register_buffer("project", torch.linspace(0, reg_max, reg_max + 1))

    def forward(self, x: Tensor) -> Tensor:
        """gfl_head.py:47-62: [n, 4 (reg_max + 1)] side distributions -> [n, 4] = softmax . (0, 1, ..., reg_max)"""
        from . import leaf
        return leaf.integral(x, self.reg_max)


# ---------------------------------------------------------------------------------------------------
# heads
# ---------------------------------------------------------------------------------------------------
@MODELS.register_module()
class GFLHead(nn.Module):
    def __init__(self, num_classes: int, in_channels: int, stacked_convs: int = 4, conv_cfg=None,
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                 loss_dfl=dict(type="DistributionFocalLoss", loss_weight=0.25),
                 bbox_coder=dict(type="DistancePointBBoxCoder"), reg_max: int = 16, init_cfg=None,
                 feat_channels: int = 256,
                 anchor_generator=dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                                       strides=[8, 16, 32, 64, 128]),
                 loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0),
                 loss_bbox=dict(type="GIoULoss", loss_weight=2.0), reg_decoded_bbox=False, train_cfg=None,
                 test_cfg=None, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError(f"GFLHead got unexpected arguments {sorted(kwargs)}")
        self.num_classes = self.cls_out_channels = num_classes
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.stacked_convs, self.conv_cfg, self.norm_cfg, self.reg_max = stacked_convs, conv_cfg, norm_cfg, reg_max
        if (in_channels, feat_channels) != (256, 256) or reg_max != 16 or norm_cfg.get("num_groups", 32) != 32:
            raise NotImplementedError("head kernels are built for 256 channels, GN(32), reg_max=16")
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.prior_generator = TASK_UTILS.build(anchor_generator)
        self.bbox_coder = TASK_UTILS.build(bbox_coder)
        self.loss_cls = MODELS.build(loss_cls)
        self.loss_bbox = MODELS.build(loss_bbox)
        self.loss_dfl = MODELS.build(loss_dfl)
        # the fused loss kernels of the training step are built for the ERD configs' settings (the modules' own
        # `forward` accepts every reduction)
        for m in (self.loss_cls, self.loss_bbox, self.loss_dfl):
            if getattr(m, "reduction", "mean") != "mean":
                raise NotImplementedError(f"{type(m).__name__}(reduction={m.reduction!r}) in a GFL head: the fused step kernels use 'mean'")
        if getattr(self.loss_bbox, "eps", 1e-6) != 1e-6:
            raise NotImplementedError("GIoULoss(eps != 1e-6) in a GFL head: the fused step kernels use the module default")
        if train_cfg:
            self.assigner = TASK_UTILS.build(train_cfg["assigner"])
            if train_cfg.get("allowed_border", -1) >= 0 or train_cfg.get("pos_weight", -1) > 0:
                raise NotImplementedError("allowed_border>=0 / pos_weight>0 are outside the ERD configs")
        self._init_layers()
        self.integral = Integral(reg_max)
        self.loss_cls_for_replay_v3 = MODELS.build(dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0))

    def _init_layers(self):
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            self.cls_convs.append(ConvModule(256, 256, 3, padding=1, norm_cfg=self.norm_cfg, init="normal"))
            self.reg_convs.append(ConvModule(256, 256, 3, padding=1, norm_cfg=self.norm_cfg, init="normal"))
        self.gfl_cls = ConvHolder(256, self.cls_out_channels, 3, 1, 1, bias=True)
        self.gfl_reg = ConvHolder(256, 4 * (self.reg_max + 1), 3, 1, 1, bias=True)
        self.gfl_cls.reset_parameters("normal", 0.01, bias=float(-math.log((1 - 0.01) / 0.01)))   # bias_prob=0.01
        self.gfl_reg.reset_parameters("normal", 0.01)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.prior_generator.strides])

    # -- forward ------------------------------------------------------------------------------------
    def forward_cat(self, p_cat: Tensor, sizes) -> Tuple[Tensor, Tensor]:
        """[N,A,256] -> (cls [N,A,C], bbox [N,A,68]) : gfl_head.py:205-230 on all levels at once."""
        _gpu_only(p_cat, "GFLHead.forward")
        sizes = [tuple(s) for s in sizes]
        # the two towers are independent: the reg tower runs on an auxiliary HIP stream (autograd replays its
        # backward on that stream too), so both directions keep two MFMA-bound kernel streams in flight
        cur = torch.cuda.current_stream(p_cat.device)
        aux = Fn.aux_stream(p_cat.device)
        forked = aux.cuda_stream != cur.cuda_stream      # (a stream never waits on itself: see functional.CAPTURE_ORIGIN)
        if forked:
            aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            r = p_cat
            for m in self.reg_convs:
                r = Fn.HeadConvGN.apply(r, m.conv.weight, m.gn.weight, m.gn.bias, sizes, m.gn.eps)
            pre = Fn.HeadConvBias.apply(r, self.gfl_reg.weight, self.gfl_reg.bias, sizes)
            alphas = torch.stack([s.scale for s in self.scales])
            bbox = Fn.LevelScale.apply(pre, alphas, sizes)
        c = p_cat
        for m in self.cls_convs:
            c = Fn.HeadConvGN.apply(c, m.conv.weight, m.gn.weight, m.gn.bias, sizes, m.gn.eps)
        cls = Fn.HeadConvBias.apply(c, self.gfl_cls.weight, self.gfl_cls.bias, sizes)
        if forked:
            cur.wait_stream(aux)
            bbox.record_stream(cur)
        return cls, bbox

    def forward(self, x: Sequence[Tensor]):
        sizes = [tuple(f.shape[-2:]) for f in x]
        cat = torch.cat([_nhwc(f).reshape(f.shape[0], -1, f.shape[1]) for f in x], 1)     # layout plumbing
        cls, bbox = self.forward_cat(cat, sizes)
        return ([_nchw(v) for v in K.level_views(cls, sizes)], [_nchw(v) for v in K.level_views(bbox, sizes)])

    # -- inference ------------------------------------------------------------
    def predict_by_feat_cat(self, cls: Tensor, bbox: Tensor, sizes, batch_img_metas, cfg=None, rescale: bool = False,
                            with_nms: bool = True) -> List[InstanceData]:
        """BaseDenseHead.predict_by_feat + GFLHead._predict_by_feat_single + _bbox_post_process
        (base_dense_head.py:201-289,424-486; gfl_head.py:408-502) for the whole batch on the GPU: two C-ABI calls
        (erd_predict_topk, erd_predict_nms), one device->host copy of the N detection counts at the end."""
        _gpu_only(cls, "GFLHead.predict_by_feat")
        cfg = self.test_cfg if cfg is None else cfg
        if cfg is None:
            raise ValueError("predict needs a test_cfg (nms_pre, score_thr, nms, max_per_img)")
        if not with_nms:
            raise NotImplementedError("with_nms=False is the test-time-augmentation path (out of scope)")
        nms = dict(cfg["nms"])
        if nms.get("type", "nms") != "nms" or nms.get("class_agnostic", False):
            raise NotImplementedError("only class-aware hard NMS (the ERD configs) is built")
        sizes = [tuple(s) for s in sizes]
        N, dev = cls.shape[0], cls.device
        anchors = self.prior_generator.grid_priors_cat(sizes, dev)
        hw = torch.tensor([[float(m["img_shape"][0]), float(m["img_shape"][1])] for m in batch_img_metas],
                          dtype=torch.float32).to(dev, non_blocking=True)
        if rescale:
            for m in batch_img_metas:
                assert m.get("scale_factor") is not None       # base_dense_head.py:459
            inv = [[1 / m["scale_factor"][0], 1 / m["scale_factor"][1]] for m in batch_img_metas]
        else:
            inv = [[1.0, 1.0]] * N
        inv = torch.tensor(inv, dtype=torch.float32).to(dev, non_blocking=True)
        nms_pre = int(cfg.get("nms_pre", -1))
        if nms_pre <= 0:
            raise NotImplementedError("nms_pre <= 0 (keep every candidate) is not built; the ERD configs use 1000")
        strides = list(self.prior_generator.strides)
        boxes, scores, labels, num = K.predict_topk(cls.contiguous(), bbox.contiguous(), anchors, sizes, strides, hw,
                                                    float(cfg["score_thr"]), nms_pre)
        dets, det_labels, det_num = K.predict_nms(boxes, scores, labels, num, inv, float(cfg.get("min_bbox_size", -1)),
                                                  float(nms["iou_threshold"]), int(cfg["max_per_img"]))
        counts = det_num.tolist()
        out = []
        for n in range(N):
            r = InstanceData()
            r.bboxes = dets[n, :counts[n], :4]
            r.scores = dets[n, :counts[n], 4]
            r.labels = det_labels[n, :counts[n]]
            out.append(r)
        return out

    def predict_by_feat(self, cls_scores: Sequence[Tensor], bbox_preds: Sequence[Tensor], score_factors=None,
                        batch_img_metas=None, cfg=None, rescale: bool = False, with_nms: bool = True):
        """reference signature (per-level NCHW maps)"""
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        cat = lambda ms: torch.cat([_nhwc(m).reshape(m.shape[0], -1, m.shape[1]) for m in ms], 1).contiguous()
        return self.predict_by_feat_cat(cat(cls_scores), cat(bbox_preds), sizes, batch_img_metas, cfg, rescale, with_nms)

    def predict(self, x: Sequence[Tensor], batch_data_samples, rescale: bool = False):
        """base_dense_head.py:171-199"""
        metas = [d.metainfo for d in batch_data_samples]
        with torch.no_grad():
            cls, bbox = self.forward(x)
        return self.predict_by_feat(cls, bbox, batch_img_metas=metas, rescale=rescale)

    # -- targets ------------------------------------------------------------------------------------
    def _targets(self, sizes, batch_gt_instances, batch_img_metas, device) -> SimpleNamespace:
        """anchors + ATSS assignment for the batch (gfl_head.py:504-669) -- all on the GPU, no host sync."""
        N = len(batch_img_metas)
        boxes = [g.bboxes for g in batch_gt_instances]
        labels = [g.labels for g in batch_gt_instances]
        counts = [int(b.shape[0]) for b in boxes]
        off = np.zeros(N + 1, dtype=np.int32)
        off[1:] = np.cumsum(counts)
        if off[-1] > 0:
            gb = torch.cat([b.reshape(-1, 4).float() for b in boxes], 0).to(device, non_blocking=True).contiguous()
            gl = torch.cat([l.reshape(-1).long() for l in labels], 0).to(device, non_blocking=True).contiguous()
        else:
            gb = torch.zeros((1, 4), device=device)
            gl = torch.zeros((1,), dtype=torch.long, device=device)
        goff = torch.from_numpy(off).to(device, non_blocking=True)
        return self._targets_packed(sizes, gb, gl, goff, max(counts + [0]), batch_img_metas, device)

    def _targets_packed(self, sizes, gb: Tensor, gl: Tensor, goff: Tensor, max_gt: int, batch_img_metas, device) -> SimpleNamespace:
        """the device half of `_targets`: packed ground truth (boxes [G,4], labels [G], per-image offsets [N+1], all on the
        device; `max_gt` >= the largest per-image count, it only sizes the assignment grid) -> anchors + ATSS targets.
        A captured training step (engine.StepGraphs) calls this with static buffers."""
        N = len(batch_img_metas)
        anchors = self.prior_generator.grid_priors_cat(sizes, device)
        valid = None
        full = all(all(min(int(np.ceil(m["pad_shape"][0] / s)), fh) == fh and min(int(np.ceil(m["pad_shape"][1] / s)), fw) == fw
                       for (fh, fw), s in zip(sizes, self.prior_generator.strides)) for m in batch_img_metas)
        if not full:
            flags = [torch.cat(self.prior_generator.valid_flags(sizes, m["pad_shape"]), 0) for m in batch_img_metas]
            for f in flags:
                if not bool(f.any()):
                    raise ValueError("There is no valid anchor inside the image boundary. Please check the image "
                                     "size and anchor sizes, or set ``allowed_border`` to -1 to skip the condition.")
            valid = torch.stack(flags).to(device, non_blocking=True)
        lab, lw, bt, npos = K.atss_assign(anchors, valid, sizes, gb, gl, goff, N, max_gt, self.num_classes, self.assigner.topk)
        return SimpleNamespace(anchors=anchors, labels=lab, label_weights=lw, bbox_targets=bt, num_pos=npos,
                               sizes=[tuple(s) for s in sizes], strides=list(self.prior_generator.strides))

    def _loss_vector(self, s_cls, s_bbox, t: SimpleNamespace) -> Tensor:
        t.lw_cls, t.lw_bbox, t.lw_dfl = self.loss_cls.loss_weight, self.loss_bbox.loss_weight, self.loss_dfl.loss_weight
        t.world_size = _world_size()
        return Fn.ERDLossFn.apply(s_cls, s_bbox, t)

    def loss_by_feat_cat(self, s_cls, s_bbox, sizes, batch_gt_instances, batch_img_metas) -> dict:
        """GFLHead.loss_by_feat (gfl_head.py:339-406) on concatenated maps."""
        t = self._targets(sizes, batch_gt_instances, batch_img_metas, s_cls.device)
        t.c_old, t.distill, t.lw_ld, t.T, t.dist_loss_weight = 0, False, 0.0, 1.0, 0.0
        v = self._loss_vector(s_cls, s_bbox, t)
        L = len(sizes)
        return LossDict(v[:3 * L], dict(loss_cls=(0, L), loss_bbox=(L, 2 * L), loss_dfl=(2 * L, 3 * L)))


@MODELS.register_module()
class GFLHeadIncrementERD(GFLHead):
    """gfl_head_increment_erd.py:57-484.  D11: like the reference, only `num_classes, in_channels, bbox_coder,
    init_cfg, **kwargs` reach GFLHead.__init__, so `stacked_convs / conv_cfg / norm_cfg / reg_max` given to
    THIS class are silently replaced by GFLHead's defaults (4 / None / GN32 / 16)."""

    def __init__(self, num_classes: int, in_channels: int, stacked_convs: int = 4, conv_cfg=None,
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True),
                 loss_dfl=dict(type="DistributionFocalLoss", loss_weight=0.25),
                 loss_ld=dict(type="KnowledgeDistillationKLDivLoss", loss_weight=0.25, T=10),
                 bbox_coder=dict(type="DistancePointBBoxCoder"), reg_max: int = 16, init_cfg=None, **kwargs):
        super().__init__(num_classes=num_classes, in_channels=in_channels, bbox_coder=bbox_coder, init_cfg=init_cfg,
                         **kwargs)
        self.loss_dfl = MODELS.build(loss_dfl)
        self.loss_ld = MODELS.build(loss_ld)

    def loss_cat(self, t_cls, t_bbox, s_cls, s_bbox, sizes, batch_data_samples, ers: dict, keep: Tensor,
                 ori_num_classes: int, dist_loss_weight: float, targets: Optional[SimpleNamespace] = None) -> dict:
        if targets is None:
            gts, _, metas = unpack_gt_instances(batch_data_samples)
            targets = self._targets(sizes, gts, metas, s_cls.device)
        t = targets
        t.c_old, t.distill = int(ori_num_classes), True
        t.t_cls, t.t_bbox, t.ers, t.keep = t_cls, t_bbox, ers, keep
        t.lw_ld, t.T, t.dist_loss_weight = self.loss_ld.loss_weight, float(self.loss_ld.T), float(dist_loss_weight)
        v = self._loss_vector(s_cls, s_bbox, t)
        L, N = len(sizes), s_cls.shape[0]
        return LossDict(v, dict(loss_cls=(0, L), loss_bbox=(L, 2 * L), loss_dfl=(2 * L, 3 * L),
                                loss_dist_cls=(3 * L, 3 * L + N), loss_dist_bbox=(3 * L + N, 3 * L + 2 * N)))

    def loss(self, ori_outs, new_outs, batch_data_samples, topk_cls_inds, topk_cls_scores, topk_bbox_inds,
             topk_bbox_preds, ori_num_classes, dist_loss_weight, model) -> dict:
        """Reference signature (gfl_head_increment_erd.py:457-459): per-level NCHW lists + index lists."""
        sizes = [tuple(m.shape[-2:]) for m in new_outs[0]]
        flat = lambda maps: torch.cat([_nhwc(m).reshape(m.shape[0], -1, m.shape[1]) for m in maps], 1).contiguous()
        t_cls, t_bbox = flat(ori_outs[0]).detach(), flat(ori_outs[1]).detach()
        s_cls, s_bbox = flat(new_outs[0]), flat(new_outs[1])
        N, A = s_cls.shape[:2]
        dev = s_cls.device
        idx_c = torch.zeros((N, A), dtype=torch.long, device=dev)
        idx_b = torch.zeros((N, A), dtype=torch.long, device=dev)
        counts = torch.zeros((N, 2), dtype=torch.int32)
        for i in range(N):
            idx_c[i, :len(topk_cls_inds[i])] = topk_cls_inds[i]
            idx_b[i, :len(topk_bbox_inds[i])] = topk_bbox_inds[i]
            counts[i, 0], counts[i, 1] = len(topk_cls_inds[i]), len(topk_bbox_inds[i])
        ers = dict(idx_cls=idx_c, idx_bbox=idx_b, counts=counts.to(dev))
        anchors = self.prior_generator.grid_priors_cat(sizes, dev)
        keep, _ = K.distill_nms(t_cls, t_bbox, anchors, idx_b, ers["counts"], 0.005)
        return self.loss_cat(t_cls, t_bbox, s_cls, s_bbox, sizes, batch_data_samples, ers, keep, ori_num_classes,
                             dist_loss_weight)


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


# ---------------------------------------------------------------------------------------------------
# detectors
# ---------------------------------------------------------------------------------------------------
@MODELS.register_module()
class DetDataPreprocessor(nn.Module):
    """data_preprocessor.py:110-183 + mmengine ImgDataPreprocessor: BGR->RGB, (x-mean)/std, zero-pad to a
    multiple of `pad_size_divisor` AFTER normalisation, stamp pad_shape / batch_input_shape.
    Input side of the step (R21); the synthetic benchmark feeds already-preprocessed tensors."""

    def __init__(self, mean=None, std=None, pad_size_divisor: int = 1, pad_value=0, bgr_to_rgb: bool = False,
                 rgb_to_bgr: bool = False, **kwargs):
        super().__init__()
        self.pad_size_divisor, self.pad_value = pad_size_divisor, pad_value
        self.channel_conversion = bgr_to_rgb or rgb_to_bgr
        self.register_buffer("mean", torch.tensor(mean if mean is not None else [0., 0., 0.]).view(3, 1, 1), False)
        self.register_buffer("std", torch.tensor(std if std is not None else [1., 1., 1.]).view(3, 1, 1), False)
        self._mean_host = [float(v) for v in self.mean.flatten()]      # fp32-rounded, as the buffers hold them
        self._std_host = [float(v) for v in self.std.flatten()]

    def forward(self, data: dict, training: bool = False) -> dict:
        imgs, samples = data["inputs"], data.get("data_samples")
        d = self.pad_size_divisor
        H = max(int(math.ceil(im.shape[1] / d)) * d for im in imgs)
        W = max(int(math.ceil(im.shape[2] / d)) * d for im in imgs)
        dev = self.mean.device
        if dev.type != "cuda":
            raise RuntimeError("DetDataPreprocessor runs on the GPU (erd_amd has no CPU path): move the model first")
        batch = torch.empty((len(imgs), 3, H, W), dtype=torch.float32, device=dev)
        mean, std = self._mean_host, self._std_host
        for i, im in enumerate(imgs):
            x = im.to(dev, non_blocking=True).contiguous()
            if x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            K.preprocess_into(x, batch[i], mean, std, self.channel_conversion, float(self.pad_value))
            if samples is not None:
                samples[i].set_metainfo(dict(img_shape=(im.shape[1], im.shape[2]), pad_shape=(H, W),
                                             batch_input_shape=(H, W)))
        return dict(inputs=batch, data_samples=samples)


@MODELS.register_module()
class GFL(nn.Module):
    """SingleStageDetector/GFL (single_stage.py:20-149, gfl.py:31-46, base.py:58-99)."""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, data_preprocessor=None,
                 init_cfg=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor) if data_preprocessor else None
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck) if neck is not None else None
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)      # single_stage.py:33-34
        self.bbox_head = MODELS.build(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._is_init = False

    @property
    def with_neck(self) -> bool:
        return self.neck is not None

    def init_weights(self) -> None:
        """mmengine BaseModule.init_weights as the reference Runner calls it before training: a no-op once the model
        is initialised (`_is_init`; GFLIncrementERD sets it after the teacher / warm start, gfl_increment_erd.py:63-65),
        otherwise the backbone's `init_cfg` ('Pretrained') is honoured; neck and head hold their init already."""
        if not self._is_init:
            self.backbone.init_weights()
            self._is_init = True

    def extract_feat(self, batch_inputs: Tensor) -> Tuple[Tensor, ...]:
        x = self.backbone(batch_inputs)
        return self.neck(x) if self.with_neck else x

    def _forward_cat(self, batch_inputs: Tensor, trunk=None):
        p_cat, sizes = self.neck.forward_cat(self.backbone(batch_inputs, trunk=trunk) if trunk is not None
                                             else self.backbone(batch_inputs))
        cls, bbox = self.bbox_head.forward_cat(p_cat, sizes)
        return cls, bbox, sizes

    def _forward(self, batch_inputs: Tensor, batch_data_samples=None):
        cls, bbox, sizes = self._forward_cat(batch_inputs)
        return ([_nchw(v) for v in K.level_views(cls, sizes)], [_nchw(v) for v in K.level_views(bbox, sizes)])

    def loss(self, batch_inputs: Tensor, batch_data_samples) -> dict:
        cls, bbox, sizes = self._forward_cat(batch_inputs)
        gts, _, metas = unpack_gt_instances(batch_data_samples)
        return self.bbox_head.loss_by_feat_cat(cls, bbox, sizes, gts, metas)

    def predict(self, batch_inputs, batch_data_samples, rescale: bool = True):
        """SingleStageDetector.predict (single_stage.py:78-112) + add_pred_to_datasample (base.py:130-156)."""
        with torch.no_grad():
            cls, bbox, sizes = self._forward_cat(batch_inputs)
            results = self.bbox_head.predict_by_feat_cat(cls, bbox, sizes, [d.metainfo for d in batch_data_samples],
                                                         rescale=rescale)
        for d, r in zip(batch_data_samples, results):
            d.pred_instances = r
        return batch_data_samples

    def forward(self, inputs: Tensor, data_samples=None, mode: str = "tensor"):
        if mode == "loss":
            return self.loss(inputs, data_samples)
        if mode == "predict":
            return self.predict(inputs, data_samples)
        if mode == "tensor":
            return self._forward(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')


def load_state_dict_strict(module: nn.Module, state_dict: Dict[str, Tensor]) -> None:
    if list(state_dict.keys()) and list(state_dict.keys())[0].startswith("module."):
        state_dict = {k[7:]: v for k, v in state_dict.items()}
    module.load_state_dict(state_dict, strict=True)


def load_checkpoint(model: nn.Module, filename: str, strict: bool = True) -> dict:
    ckpt = torch.load(filename, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if not isinstance(sd, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    load_state_dict_strict(model, sd)
    return ckpt


@MODELS.register_module()
class GFLIncrementERD(GFL):
    """gfl_increment_erd.py:20-220."""

    def __init__(self, ori_setting, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, data_preprocessor=None,
                 init_cfg=None, latest_model_flag=True, top_k=100, dist_loss_weight=1):
        super().__init__(backbone=backbone, neck=neck, bbox_head=bbox_head, train_cfg=train_cfg, test_cfg=test_cfg,
                         data_preprocessor=data_preprocessor, init_cfg=init_cfg)
        self.top_k = top_k                      # stored, never read (D2)
        self.dist_loss_weight = dist_loss_weight
        self.teacher_stream: Optional[torch.cuda.Stream] = None
        if latest_model_flag:
            self.load_base_detector(ori_setting)
            self._is_init = True

    def _load_checkpoint_for_new_model(self, checkpoint_file: str, strict: bool = True) -> None:
        """:67-93 -- warm start: teacher checkpoint with gfl_cls widened by the student's fresh new-class rows."""
        checkpoint = torch.load(checkpoint_file, map_location="cpu", weights_only=False)
        if isinstance(checkpoint, OrderedDict):
            state_dict = checkpoint
        elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
            state_dict = checkpoint["state_dict"]
        else:
            raise RuntimeError("No state_dict found in checkpoint file {}".format(checkpoint_file))
        if list(state_dict.keys())[0].startswith("module."):
            state_dict = {k[7:]: v for k, v in checkpoint["state_dict"].items()}
        state_dict = dict(state_dict)
        w, b = self.bbox_head.gfl_cls.weight.detach().cpu(), self.bbox_head.gfl_cls.bias.detach().cpu()
        state_dict["bbox_head.gfl_cls.weight"] = torch.cat(
            (state_dict["bbox_head.gfl_cls.weight"], w[self.ori_num_classes:, ...]), dim=0)
        state_dict["bbox_head.gfl_cls.bias"] = torch.cat(
            (state_dict["bbox_head.gfl_cls.bias"], b[self.ori_num_classes:, ...]), dim=0)
        self.load_state_dict(state_dict, strict=strict)

    def load_base_detector(self, ori_setting) -> None:
        """:95-122"""
        assert os.path.isfile(ori_setting["ori_checkpoint_file"]), "{} is not a valid file".format(
            ori_setting["ori_checkpoint_file"])
        ori_cfg = Config.fromfile(ori_setting["ori_config_file"])
        if "latest_model_flag" in ori_cfg.model:
            ori_cfg.model.latest_model_flag = False
        ori_model = MODELS.build(ori_cfg.model)
        load_checkpoint(ori_model, ori_setting["ori_checkpoint_file"], strict=True)
        self.ori_num_classes = ori_setting["ori_num_classes"]
        self._load_checkpoint_for_new_model(ori_setting["ori_checkpoint_file"])
        print("======> load base checkpoint for new model from {}".format(ori_setting["ori_checkpoint_file"]), file=sys.stderr)
        self.attach_teacher(ori_model, self.ori_num_classes)

    def attach_teacher(self, ori_model: nn.Module, ori_num_classes: int) -> None:
        ori_model.eval()
        for p in ori_model.parameters():
            p.requires_grad = False
        self.ori_num_classes = ori_num_classes
        self.ori_model = ori_model      # registered submodule, as in the reference (D7)

    # -- ERS -------------------------------------------------------------------------------------------
    def sel_pos_cat(self, t_cls: Tensor, t_bbox: Tensor) -> dict:
        """:143-200 on [N,A,C] buffers; everything stays on the device (no nonzero() sync)."""
        return K.ers_select(t_cls, t_bbox)

    def sel_pos(self, cls_scores: List[Tensor], bbox_preds: List[Tensor]):
        """reference-shaped API: per-image index lists + gathered rows (forces a host sync for the counts)."""
        flat = lambda maps: torch.cat([_nhwc(m).reshape(m.shape[0], -1, m.shape[1]) for m in maps], 1).contiguous()
        tc, tb = flat(cls_scores), flat(bbox_preds)
        r = self.sel_pos_cat(tc, tb)
        cnt = r["counts"].cpu()
        ic = [r["idx_cls"][i, :int(cnt[i, 0])] for i in range(tc.shape[0])]
        ib = [r["idx_bbox"][i, :int(cnt[i, 1])] for i in range(tc.shape[0])]
        return ic, [tc[i][ic[i]] for i in range(len(ic))], ib, [tb[i][ib[i]] for i in range(len(ib))]

    def shares_trunk(self) -> bool:
        """True when the student's frozen front (stem + the first `frozen_stages` stages) is bit-identical to the
        teacher's: same depth / frozen_stages, no trainable tensor in it, every parameter and buffer equal.  Checked on
        the device once per (storage, version) state -- loading other weights into either network re-checks."""
        if os.environ.get("ERD_SHARE_TRUNK", "1") == "0":
            return False
        a, b = self.backbone, self.ori_model.backbone
        if a.depth != b.depth or a.frozen_stages < 1 or a.frozen_stages != b.frozen_stages:
            return False
        ta, tb = a.trunk_tensors(), b.trunk_tensors()
        key = tuple((t.data_ptr(), t._version) for t in ta + tb)
        hit = getattr(self, "_trunk_shared", None)
        if hit is None or hit[0] != key:
            same = len(ta) == len(tb) and not any(t.requires_grad for t in ta) and all(
                x.shape == y.shape and bool(torch.equal(x, y)) for x, y in zip(ta, tb))
            hit = self._trunk_shared = (key, same)
        return hit[1]

    def teacher_pass(self, batch_inputs: Tensor, batch_data_samples=None, share_trunk: bool = True) -> "TeacherOut":
        """the no-grad half of `loss` (:205-208): teacher forward, ERS, the NMS of the selected teacher boxes and
        (when the data samples are given) the ATSS targets -- everything that does not depend on the student's
        parameters, so the trainer runs it on a side stream."""
        trunk, trunk_event = None, None
        if self.shares_trunk():
            # student and teacher hold the same frozen stem + layer1 (the warm start copies them, :83-93, and nothing
            # updates them): computed ONCE, on the kernels the student's trunk uses, and fed to both networks
            with K.distillation_forward(K.WINO_FROZEN_TRUNK):
                trunk = self.backbone.trunk(batch_inputs)
            # share_trunk=False: same arithmetic, but the map is not handed on.  "static": handed on without an event --
            # a hipGraph capture (engine.TeacherGraphs records the hand-over event after each replay)
            if share_trunk is True:
                trunk_event = torch.cuda.Event()
                trunk_event.record()
        with K.distillation_forward(K.WINO_TEACHER):
            t_cls, t_bbox, sizes = self.ori_model._forward_cat(batch_inputs, trunk=trunk)
        ers = self.sel_pos_cat(t_cls, t_bbox)
        anchors = self.bbox_head.prior_generator.grid_priors_cat(sizes, t_cls.device)
        keep, kcnt = K.distill_nms(t_cls, t_bbox, anchors, ers["idx_bbox"], ers["counts"], 0.005)
        out = TeacherOut(t_cls, t_bbox, sizes, ers, keep, kcnt)
        out.trunk, out.trunk_event = (trunk, trunk_event) if share_trunk else (None, None)
        if batch_data_samples is not None:
            gts, _, metas = unpack_gt_instances(batch_data_samples)
            out.targets = self.bbox_head._targets(sizes, gts, metas, t_cls.device)
        return out

    def loss(self, batch_inputs: Tensor, batch_data_samples, teacher_out: Optional["TeacherOut"] = None) -> dict:
        """:202-220: teacher fwd -> ERS (+ NMS) -> student fwd -> losses."""
        if teacher_out is None:
            with torch.no_grad():   # D6: the reference omits no_grad; teacher params are frozen => identical
                teacher_out = self.teacher_pass(batch_inputs)
        t = teacher_out
        with K.distillation_forward(K.WINO_FROZEN_TRUNK):
            s_cls, s_bbox, sizes = self._forward_cat(batch_inputs, trunk=t.trunk)
        return self.bbox_head.loss_cat(t.t_cls, t.t_bbox, s_cls, s_bbox, sizes, batch_data_samples, t.ers, t.keep,
                                       self.ori_num_classes, self.dist_loss_weight, targets=t.targets)


class TeacherOut:
    def __init__(self, t_cls, t_bbox, sizes, ers, keep, keep_count):
        self.t_cls, self.t_bbox, self.sizes, self.ers, self.keep, self.keep_count = t_cls, t_bbox, sizes, ers, keep, keep_count
        self.targets: Optional[SimpleNamespace] = None
        self.trunk = None             # (h, outs) of the shared frozen trunk, or None: the student computes its own
        self.trunk_event = None       # recorded on the producing stream right after the trunk

    def tensors(self) -> List[Tensor]:
        out = [self.t_cls, self.t_bbox, self.keep, self.keep_count] + list(self.ers.values())
        if self.trunk is not None:
            out += [self.trunk[0]] + list(self.trunk[1])
        if self.targets is not None:
            out += [v for v in vars(self.targets).values() if isinstance(v, torch.Tensor)]
        return out


class LossDict(dict):
    """The reference's loss dict (name -> list of 0-dim tensors), built from the ONE vector the loss kernels emit.
    `parse_losses` recognises it and sums the vector directly: a handful of launches instead of ~110 (a mean per
    entry, Python `sum` chains and their select/add backward nodes)."""

    def __init__(self, vector: Tensor, slices: Dict[str, Tuple[int, int]]):
        super().__init__({k: list(vector[a:b].unbind(0)) for k, (a, b) in slices.items()})
        self.vector, self.slices = vector, dict(slices)

    def intact(self) -> bool:
        return list(self.keys()) == list(self.slices.keys()) and all(
            isinstance(v, list) and len(v) == b - a for v, (a, b) in zip(self.values(), self.slices.values()))


def parse_losses(losses: Dict[str, Union[Tensor, List[Tensor]]]) -> Tuple[Tensor, Dict[str, Tensor]]:
    """mmengine BaseModel.parse_losses (D9): tensor -> mean, list -> sum of means; total over keys with 'loss'."""
    log_vars = OrderedDict()
    if isinstance(losses, LossDict) and losses.intact() and all("loss" in k for k in losses.slices):
        d = losses.vector.detach()
        for name, (a, b) in losses.slices.items():
            log_vars[name] = d[a:b].sum()
        total = losses.vector.sum()
        log_vars["loss"] = total.detach()      # (the returned `total` carries the graph; a logged one would pin it)
        return total, log_vars
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f"{name} is not a tensor or list of tensors")
    total = sum(v for k, v in log_vars.items() if "loss" in k)
    log_vars = OrderedDict((k, v.detach()) for k, v in log_vars.items())
    log_vars["loss"] = total.detach()
    return total, log_vars
