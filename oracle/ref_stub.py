"""TEST INFRASTRUCTURE ONLY -- build-container helper, never imported by the product.

Imports the reference's hot-path Python files *where they lie* under
``/root/reference`` (read-only) without running any package ``__init__`` and
without mmcv / mmengine (absent from this image), following the recipe of
SURVEY.md section 8(c).  It exists for two purposes only:

  1. ``oracle/gen_golden.py`` uses it to generate the committed fixtures under
     ``tests/golden/``;
  2. ``tests/test_oracle_vs_reference.py`` uses it (when ``/root/reference``
     is present, i.e. in the build container only) to validate the restatement
     in ``oracle/erd_oracle.py`` against the real reference source.

Everything defined here is a *stand-in for the un-vendored third-party
packages* (mmengine==0.7.3, mmcv==2.0.0: thin wrappers over torch whose
semantics are fully determined by torch -- see SURVEY.md 8(c)); no reference
source text is copied.  ``mmcv.ops.batched_nms`` is restated from its published
algorithm (class-offset trick + greedy IoU>thr suppression, score-descending
keep order) and is flagged *unpinned vs. mmcv* everywhere it is used.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("ERD_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "mmdet", "models", "detectors", "gfl_increment_erd.py"))


# ----------------------------------------------------------------------------
# third-party stand-ins
# ----------------------------------------------------------------------------
class ConfigDict(dict):
    """attr-dict (mmengine.config.ConfigDict behaviour used by the path)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            self[key] = _wrap(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def copy(self):
        return ConfigDict(super().copy())

    def __deepcopy__(self, memo):
        import copy as _c
        return ConfigDict({k: _c.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class InstanceData:
    """attribute bag with __contains__ (mmengine.structures.InstanceData subset)."""

    def __init__(self, metainfo=None, **kwargs):
        object.__setattr__(self, "_fields", {})
        object.__setattr__(self, "_meta", dict(metainfo or {}))
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        f = object.__getattribute__(self, "_fields")
        if k in f:
            return f[k]
        m = object.__getattribute__(self, "_meta")
        if k in m:
            return m[k]
        raise AttributeError(k)

    def __contains__(self, k):
        return k in self._fields or k in self._meta

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def keys(self):
        return list(self._fields.keys())

    def __getitem__(self, item):
        """index every field alike (mmengine.structures.InstanceData.__getitem__: mask / index tensor / slice)"""
        out = InstanceData(metainfo=self._meta)
        for k, v in self._fields.items():
            setattr(out, k, v[item])
        return out

    def pop(self, k, *default):
        return self._fields.pop(k, *default)

    @property
    def metainfo(self):
        return dict(self._meta)


class DetDataSample:
    def __init__(self, metainfo=None):
        self._meta = dict(metainfo or {})
        self.gt_instances = None
        self.ignored_instances = None

    @property
    def metainfo(self):
        return dict(self._meta)

    def __contains__(self, k):
        return getattr(self, k, None) is not None


class Registry:
    def __init__(self, name):
        self.name = name
        self._m = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._m[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._m.get(key)

    def build(self, cfg, default_args=None, **kw):
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        t = cfg.pop("type")
        cls = self._m[t] if isinstance(t, str) else t
        return cls(**cfg)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self._is_init = False

    def init_weights(self):
        pass


class BaseModel(BaseModule):
    def __init__(self, data_preprocessor=None, init_cfg=None):
        super().__init__(init_cfg)
        self.data_preprocessor = None


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule subset: conv (bias iff no norm) -> norm('gn'/'bn') -> ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, bias="auto", conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                              dilation, groups, bias=bias)
        if self.with_norm:
            t = norm_cfg["type"]
            if t == "GN":
                self.norm_name = "gn"
                norm = nn.GroupNorm(norm_cfg["num_groups"], out_channels)
            elif t == "BN":
                self.norm_name = "bn"
                norm = nn.BatchNorm2d(out_channels)
            else:
                raise NotImplementedError(t)
            self.add_module(self.norm_name, norm)
            for p in norm.parameters():
                p.requires_grad = norm_cfg.get("requires_grad", True)
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get("type", "Conv2d") in ("Conv2d", "Conv")
    return nn.Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if t == "BN":
        layer = nn.BatchNorm2d(num_features, **cfg)
        name = "bn" + str(postfix)
    elif t == "GN":
        layer = nn.GroupNorm(num_channels=num_features, **cfg)
        name = "gn" + str(postfix)
    else:
        raise NotImplementedError(t)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


def nms_restated(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """Greedy NMS, restated from mmcv 2.0.0's published algorithm (UNPINNED vs. mmcv):
    visit boxes by descending score (stable), suppress later boxes with IoU > thr
    (offset=0 areas), return kept indices in visit order."""
    n = boxes.shape[0]
    if n == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    x1, y1, x2, y2 = b.unbind(1)
    area = (x2 - x1) * (y2 - y1)
    supp = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if supp[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(x1[i], x1[i + 1:])
        yy1 = torch.maximum(y1[i], y1[i + 1:])
        xx2 = torch.minimum(x2[i], x2[i + 1:])
        yy2 = torch.minimum(y2[i], y2[i + 1:])
        w = (xx2 - xx1).clamp(min=0)
        h = (yy2 - yy1).clamp(min=0)
        inter = w * h
        iou = inter / (area[i] + area[i + 1:] - inter)
        supp[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.long)]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms restated (UNPINNED vs. mmcv): offsets idxs*(max+1) added in
    the boxes' dtype, plain NMS, output (dets[k,5], keep[k]) sorted by score desc."""
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop("class_agnostic", class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_cfg_.pop("type", "nms")
    split_thr = nms_cfg_.pop("split_thr", 10000)
    assert boxes_for_nms.shape[0] < split_thr
    keep = nms_restated(boxes_for_nms, scores, nms_cfg_["iou_threshold"])
    boxes = boxes[keep]
    scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


_LOADED = {}


def _load(modname: str, relpath: str, export_to=()):
    """exec a real reference file as module `modname`; copy its public names up."""
    if modname in _LOADED:
        return _LOADED[modname]
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    parent, _, child = modname.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], child, mod)
    for pk in export_to:
        p = sys.modules[pk]
        for k, v in mod.__dict__.items():
            cur = p.__dict__.get(k)
            if not k.startswith("_") and (cur is None or isinstance(cur, types.ModuleType)):
                if not isinstance(v, types.ModuleType):
                    setattr(p, k, v)
    _LOADED[modname] = mod
    return mod


_REF = None


def load_reference():
    """Returns a namespace with the reference's real classes/functions."""
    global _REF
    if _REF is not None:
        return _REF
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)

    # ---- third-party stand-ins -------------------------------------------------
    def digit_version(v, length=4):
        out = []
        for p in str(v).split("+")[0].split("."):
            num = "".join(ch for ch in p if ch.isdigit())
            out.append(int(num) if num else 0)
        return tuple(out + [0] * (length - len(out)))

    def is_seq_of(seq, expected_type, seq_type=None):
        if seq_type is not None and not isinstance(seq, seq_type):
            return False
        return all(isinstance(x, expected_type) for x in seq)

    def is_tuple_of(seq, expected_type):
        return is_seq_of(seq, expected_type, seq_type=tuple)

    class _Config(ConfigDict):
        @staticmethod
        def fromfile(path):
            raise RuntimeError("Config.fromfile is not used by the oracle harness")

    _pkg("mmengine")
    _mod("mmengine.config", ConfigDict=ConfigDict, Config=_Config)
    sys.modules["mmengine"].Config = _Config
    sys.modules["mmengine"].ConfigDict = ConfigDict
    _mod("mmengine.structures", InstanceData=InstanceData, PixelData=InstanceData,
         BaseDataElement=InstanceData)
    _mod("mmengine.model", BaseModule=BaseModule, BaseModel=BaseModel, Sequential=Sequential,
         constant_init=lambda *a, **k: None, xavier_init=lambda *a, **k: None,
         bias_init_with_prob=lambda p: float(-torch.log(torch.tensor((1 - p) / p))),
         normal_init=lambda *a, **k: None)
    _mod("mmengine.utils", digit_version=digit_version, is_seq_of=is_seq_of,
         is_tuple_of=is_tuple_of)
    _mod("mmengine.registry", MODELS=Registry("root_models"), Registry=Registry)
    _pkg("mmengine.runner")
    _mod("mmengine.runner.checkpoint", load_checkpoint=None, load_state_dict=None)
    _mod("mmengine.dist", get_dist_info=lambda: (0, 1))
    _mod("mmengine.logging", print_log=lambda *a, **k: None)
    _pkg("mmcv")
    _mod("mmcv.cnn", ConvModule=ConvModule, Scale=Scale, build_conv_layer=build_conv_layer,
         build_norm_layer=build_norm_layer, build_plugin_layer=None)
    _mod("mmcv.ops", batched_nms=batched_nms)
    if "cv2" not in sys.modules:
        _mod("cv2")
    if "six" not in sys.modules:
        import six  # noqa: F401  (installed)

    # ---- package shells (no reference __init__ runs) --------------------------------
    R = os.path.join(REF_ROOT, "mmdet")
    for p in ["mmdet", "mmdet.models", "mmdet.models.dense_heads", "mmdet.models.detectors",
              "mmdet.models.backbones", "mmdet.models.layers", "mmdet.models.necks",
              "mmdet.models.losses", "mmdet.models.utils", "mmdet.models.task_modules",
              "mmdet.models.task_modules.assigners", "mmdet.models.task_modules.samplers",
              "mmdet.models.task_modules.prior_generators", "mmdet.models.task_modules.coders",
              "mmdet.models.test_time_augs", "mmdet.structures", "mmdet.structures.bbox",
              "mmdet.structures.mask", "mmdet.utils"]:
        _pkg(p)
    MODELS, TASK_UTILS = Registry("model"), Registry("task util")
    _mod("mmdet.registry", MODELS=MODELS, TASK_UTILS=TASK_UTILS)
    sys.modules["mmdet"].registry = sys.modules["mmdet.registry"]

    class _DummyMask:
        pass

    _mod("mmdet.structures.mask.structures", BitmapMasks=_DummyMask, PolygonMasks=_DummyMask)
    sys.modules["mmdet.structures.mask"].BitmapMasks = _DummyMask
    sys.modules["mmdet.structures.mask"].PolygonMasks = _DummyMask
    sys.modules["mmdet.models.test_time_augs"].merge_aug_results = None
    st = sys.modules["mmdet.structures"]
    st.DetDataSample = DetDataSample
    st.SampleList = list
    st.OptSampleList = list

    @MODELS.register_module()
    class CrossEntropyLoss(nn.Module):  # built (never called) by gfl_head.py:151
        def __init__(self, **kw):
            super().__init__()

    # ---- real reference files ----------------------------------------------------
    B = "mmdet.structures.bbox"
    _load(B + ".base_boxes", "mmdet/structures/bbox/base_boxes.py", [B])
    _load(B + ".bbox_overlaps", "mmdet/structures/bbox/bbox_overlaps.py", [B])
    _load(B + ".box_type", "mmdet/structures/bbox/box_type.py", [B])
    _load(B + ".horizontal_boxes", "mmdet/structures/bbox/horizontal_boxes.py", [B])
    _load(B + ".transforms", "mmdet/structures/bbox/transforms.py", [B])
    U = "mmdet.utils"
    _load(U + ".typing_utils", "mmdet/utils/typing_utils.py", [U])
    um = _load(U + ".util_mixins", "mmdet/utils/util_mixins.py", [])
    sys.modules[U].util_mixins = um
    _load(U + ".util_random", "mmdet/utils/util_random.py", [U])
    _load(U + ".dist_utils", "mmdet/utils/dist_utils.py", [U])
    MU = "mmdet.models.utils"
    _load(MU + ".misc", "mmdet/models/utils/misc.py", [MU])
    T = "mmdet.models.task_modules"
    A = T + ".assigners"
    _load(A + ".assign_result", "mmdet/models/task_modules/assigners/assign_result.py", [A, T])
    _load(A + ".base_assigner", "mmdet/models/task_modules/assigners/base_assigner.py", [A, T])
    _load(A + ".iou2d_calculator", "mmdet/models/task_modules/assigners/iou2d_calculator.py", [A, T])
    _load(A + ".atss_assigner", "mmdet/models/task_modules/assigners/atss_assigner.py", [A, T])
    S = T + ".samplers"
    _load(S + ".sampling_result", "mmdet/models/task_modules/samplers/sampling_result.py", [S, T])
    _load(S + ".base_sampler", "mmdet/models/task_modules/samplers/base_sampler.py", [S, T])
    _load(S + ".pseudo_sampler", "mmdet/models/task_modules/samplers/pseudo_sampler.py", [S, T])
    P = T + ".prior_generators"
    _load(P + ".utils", "mmdet/models/task_modules/prior_generators/utils.py", [P, T])
    _load(P + ".anchor_generator", "mmdet/models/task_modules/prior_generators/anchor_generator.py", [P, T])
    C = T + ".coders"
    _load(C + ".base_bbox_coder", "mmdet/models/task_modules/coders/base_bbox_coder.py", [C, T])
    _load(C + ".distance_point_bbox_coder", "mmdet/models/task_modules/coders/distance_point_bbox_coder.py", [C, T])
    L = "mmdet.models.losses"
    _load(L + ".utils", "mmdet/models/losses/utils.py", [L])
    _load(L + ".gfocal_loss", "mmdet/models/losses/gfocal_loss.py", [L])
    _load(L + ".kd_loss", "mmdet/models/losses/kd_loss.py", [L])
    _load(L + ".iou_loss", "mmdet/models/losses/iou_loss.py", [L])
    D = "mmdet.models.dense_heads"
    _load(D + ".base_dense_head", "mmdet/models/dense_heads/base_dense_head.py", [D])
    _load(D + ".anchor_head", "mmdet/models/dense_heads/anchor_head.py", [D])
    _load(D + ".gfl_head", "mmdet/models/dense_heads/gfl_head.py", [D])
    _load(D + ".gfl_head_increment_erd", "mmdet/models/dense_heads/gfl_head_increment_erd.py", [D])
    LY = "mmdet.models.layers"
    _load(LY + ".res_layer", "mmdet/models/layers/res_layer.py", [LY])
    BB = "mmdet.models.backbones"
    _load(BB + ".resnet", "mmdet/models/backbones/resnet.py", [BB])
    NK = "mmdet.models.necks"
    _load(NK + ".fpn", "mmdet/models/necks/fpn.py", [NK])
    DT = "mmdet.models.detectors"
    _load(DT + ".base", "mmdet/models/detectors/base.py", [DT])
    _load(DT + ".single_stage", "mmdet/models/detectors/single_stage.py", [DT])
    _load(DT + ".gfl", "mmdet/models/detectors/gfl.py", [DT])
    _load(DT + ".gfl_increment_erd", "mmdet/models/detectors/gfl_increment_erd.py", [DT])

    ns = types.SimpleNamespace(MODELS=MODELS, TASK_UTILS=TASK_UTILS, ConfigDict=ConfigDict,
                               InstanceData=InstanceData, DetDataSample=DetDataSample,
                               batched_nms=batched_nms, nms_restated=nms_restated,
                               modules=dict(_LOADED))
    for m in _LOADED.values():
        for k, v in m.__dict__.items():
            if not k.startswith("_") and not hasattr(ns, k):
                setattr(ns, k, v)
    _REF = ns
    return ns


# ----------------------------------------------------------------------------
# reference model construction from config-shaped dicts (no Config.fromfile)
# ----------------------------------------------------------------------------
def gfl_model_cfg(num_classes: int, depth: int = 50, head_type: str = "GFLHead",
                  with_ld: bool = False):
    """The `model=` dict of configs/gfl_increment/*.py (values restated, not read)."""
    head = dict(
        type=head_type, num_classes=num_classes, in_channels=256, stacked_convs=4,
        feat_channels=256,
        anchor_generator=dict(type="AnchorGenerator", ratios=[1.0], octave_base_scale=8,
                              scales_per_octave=1, strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_dfl=dict(type="DistributionFocalLoss", loss_weight=0.25),
        reg_max=16,
        loss_bbox=dict(type="GIoULoss", loss_weight=2.0))
    if with_ld:
        head["loss_ld"] = dict(type="KnowledgeDistillationKLDivLoss", loss_weight=0.25, T=10)
    return dict(
        backbone=dict(type="ResNet", depth=depth, num_stages=4, out_indices=(0, 1, 2, 3),
                      frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=True),
                      norm_eval=True, style="pytorch", init_cfg=None),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256,
                  start_level=1, add_extra_convs="on_output", num_outs=5),
        bbox_head=head,
        train_cfg=dict(assigner=dict(type="ATSSAssigner", topk=9), allowed_border=-1,
                       pos_weight=-1, debug=False),
        test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05,
                      nms=dict(type="nms", iou_threshold=0.6), max_per_img=100))


def build_reference_erd(ori_num_classes=40, num_classes=80, depth=50, dist_loss_weight=1):
    """(teacher GFL, student GFLIncrementERD) built from the REAL reference classes.
    latest_model_flag=False; the caller loads weights and attaches the teacher
    (same as gfl_increment_erd.py:112-122 without the checkpoint file I/O)."""
    ref = load_reference()
    tcfg = ConfigDict(gfl_model_cfg(ori_num_classes, depth, "GFLHead"))
    teacher = ref.GFL(**tcfg)
    scfg = ConfigDict(gfl_model_cfg(num_classes, depth, "GFLHeadIncrementERD", with_ld=True))
    student = ref.GFLIncrementERD(ori_setting=ConfigDict(ori_num_classes=ori_num_classes),
                                  latest_model_flag=False, dist_loss_weight=dist_loss_weight,
                                  **scfg)
    return teacher, student


def attach_teacher(student, teacher, ori_num_classes):
    teacher.eval()
    for p in teacher.parameters():
        p.requires_grad = False
    student.ori_num_classes = ori_num_classes
    student.ori_model = teacher
    return student
