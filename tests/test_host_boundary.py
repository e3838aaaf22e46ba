"""CPU: the drop-in boundary -- config files, registry, state-dict ABI, teacher load /
student warm start, C-ABI exports.  No compute: erd_amd has no CPU path (and says so)."""
import os
import re
import tempfile

import pytest
import torch

import erd_amd
from erd_amd import Config, MODELS, TASK_UTILS
from e2e_util import CFG_FIRST, CFG_INCRE, ROOT
from oracle import erd_oracle as O

REF_CFG = "/root/reference/configs/gfl_increment"


def test_configs_load_and_inherit_bases():
    cfg = Config.fromfile(CFG_INCRE)
    assert cfg.model.type == "GFLIncrementERD" and cfg.model.bbox_head.type == "GFLHeadIncrementERD"
    assert cfg.model.ori_setting.ori_num_classes == 40            # attribute access on nested dicts
    assert cfg.optim_wrapper.optimizer == dict(type="SGD", lr=0.01, momentum=0.9, weight_decay=0.0001)
    assert cfg.param_scheduler[0].type == "LinearLR" and cfg.param_scheduler[1].milestones == [8, 11]   # from _base_
    assert cfg.env_cfg.dist_cfg.backend == "nccl" and cfg.auto_scale_lr == dict(enable=True, base_batch_size=16)
    assert cfg.train_dataloader.batch_size == 2                    # base value survives the child's partial override
    assert cfg.train_dataloader.dataset.ann_file.endswith("sel_last_40_cats.json")


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("name", ["gfl_r50_fpn_1x_coco_first_40_cats.py",
                                  "gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py"])
def test_reference_config_files_load_unchanged_and_equal_ours(name):
    ref = Config.fromfile(os.path.join(REF_CFG, name)).to_dict()
    ours = Config.fromfile(os.path.join(ROOT, "configs", "gfl_increment", name)).to_dict()
    ref.pop("filename"); ours.pop("filename")
    assert ref == ours
    MODELS.build(dict(ref["model"], latest_model_flag=False) if "ori_setting" in ref["model"] else ref["model"])


def test_registry_surface():
    for t in ["GFLIncrementERD", "GFL", "ResNet", "FPN", "GFLHead", "GFLHeadIncrementERD", "QualityFocalLoss",
              "DistributionFocalLoss", "GIoULoss", "KnowledgeDistillationKLDivLoss", "DetDataPreprocessor"]:
        assert t in MODELS, t
    for t in ["AnchorGenerator", "ATSSAssigner", "DistancePointBBoxCoder", "BboxOverlaps2D"]:
        assert t in TASK_UTILS, t
    with pytest.raises(KeyError):
        MODELS.build(dict(type="FasterRCNN"))
    with pytest.raises(NotImplementedError):      # out-of-path variants fail loudly instead of silently differing
        MODELS.build(dict(type="ResNet", depth=50, norm_eval=False))


def test_state_dict_abi_matches_reference_keys():
    teacher = MODELS.build(Config.fromfile(CFG_FIRST).model)
    spec = O.gfl_param_shapes(40)
    sd = teacher.state_dict()
    assert list(sorted(sd)) == list(sorted(spec))
    assert all(tuple(sd[k].shape) == spec[k] for k in spec)
    assert sum(p.numel() for p in teacher.parameters()) == 32348337                  # SURVEY 8(b) [probe]
    cfg = Config.fromfile(CFG_INCRE)
    cfg.model.latest_model_flag = False
    student = MODELS.build(cfg.model)
    assert sum(p.numel() for p in student.parameters() if p.requires_grad) == 32215193
    assert sum(p.numel() for p in student.parameters() if not p.requires_grad) == 225344
    # D11: stacked_convs / reg_max given to the ERD head are ignored, exactly like the reference
    cfg.model.bbox_head.stacked_convs = 1
    s2 = MODELS.build(cfg.model)
    assert len(s2.bbox_head.cls_convs) == 4 and s2.bbox_head.gfl_reg.weight.shape[0] == 68
    # conv weights are stored [O][kh][kw][I] while exposing the OIHW checkpoint shape
    w = student.backbone.layer2[0].conv2.weight
    assert tuple(w.shape) == (128, 128, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()


def test_teacher_load_and_student_warm_start_through_checkpoint_file():
    """gfl_increment_erd.py:67-122 with a real file on disk (latest_model_flag=True path)."""
    tsd = O.procedural_state_dict(40, seed=0)
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "epoch_12.pth")
        torch.save(dict(state_dict={"module." + k: v for k, v in tsd.items()}, meta=dict(epoch=12)), ck)
        cfg = Config.fromfile(CFG_INCRE)
        cfg.model.ori_setting.ori_checkpoint_file = ck
        cfg.model.ori_setting.ori_config_file = CFG_FIRST
        torch.manual_seed(0)
        model = MODELS.build(cfg.model)
        missing = dict(cfg.model.ori_setting, ori_checkpoint_file=os.path.join(d, "nope.pth"))
        with pytest.raises(AssertionError):
            MODELS.build(dict(cfg.model, ori_setting=missing))
    sd = model.state_dict()
    assert any(k.startswith("ori_model.") for k in sd)                  # D7: teacher is a registered submodule
    assert all(not p.requires_grad for p in model.ori_model.parameters()) and not model.ori_model.training
    assert torch.equal(sd["ori_model.bbox_head.gfl_cls.weight"], tsd["bbox_head.gfl_cls.weight"])
    assert torch.equal(sd["bbox_head.gfl_cls.weight"][:40], tsd["bbox_head.gfl_cls.weight"])     # old rows copied
    assert float(sd["bbox_head.gfl_cls.weight"][40:].std()) == pytest.approx(0.01, rel=0.1)        # new rows fresh
    assert torch.allclose(sd["bbox_head.gfl_cls.bias"][40:], torch.full((40,), -4.59512))
    assert torch.equal(sd["backbone.layer3.2.conv2.weight"], tsd["backbone.layer3.2.conv2.weight"])
    assert model.ori_num_classes == 40 and model._is_init


def test_product_has_no_cpu_path_and_never_imports_the_oracle():
    model = MODELS.build(Config.fromfile(CFG_FIRST).model)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 64, 64), mode="tensor")
    with pytest.raises(RuntimeError, match="Invalid mode"):
        model(torch.zeros(1, 3, 64, 64), mode="bogus")
    for root, _, files in os.walk(os.path.join(ROOT, "erd_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from erd_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "erd_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|size_t|const char\*)\s+(erd_\w+)\s*\(", header, re.M)))
    assert declared, "no declarations parsed"
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.erd_abi_version() == 6
    assert lib.erd_probe_build() == 0          # the shipped library contains no timing / accuracy / trace variant (csrc/erd_probes.h)
    # argument errors come back as codes + message, never as exceptions across the ABI
    assert lib.erd_conv_igemm(None, None) == -1 and b"null" in lib.erd_last_error()


def test_ctypes_signatures_match_the_header_argument_for_argument():
    """Every prototype of include/erd_hip.h against the ctypes binding the product uses (erd_amd/_lib.py): same number of
    parameters, pointers bound as pointers, 64-bit integers as 64-bit, floats as floats.  An argument added on one side only
    (the map_type parameters of ABI version 2 were such a change) would otherwise shift every later argument silently."""
    import ctypes as C
    from erd_amd import _lib
    header = open(os.path.join(ROOT, "include", "erd_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)                 # comments carry commas and parentheses
    protos = dict(re.findall(r"^(?:int|size_t)\s+(erd_\w+)\s*\(([^;]*?)\)\s*;", header, re.M | re.S))
    assert set(protos) >= set(_lib._SIGNATURES), sorted(set(_lib._SIGNATURES) - set(protos))
    for name, argtypes in _lib._SIGNATURES.items():
        params = [p.strip() for p in protos[name].split(",")] if protos[name].strip() not in ("", "void") else []
        assert len(params) == len(argtypes), (name, params, argtypes)
        for prm, ct in zip(params, argtypes):
            is_ptr = "*" in prm or prm.startswith("erd_stream_t")
            bound_ptr = ct is C.c_void_p or (hasattr(ct, "_type_") and not isinstance(ct._type_, str))
            assert is_ptr == bound_ptr, (name, prm, ct)
            if not is_ptr:
                base = prm.split()[0] if not prm.startswith("const") else prm.split()[1]
                want = {"int": C.c_int32, "int64_t": C.c_int64, "float": C.c_float, "double": C.c_double, "size_t": C.c_size_t}[base]
                assert C.sizeof(ct) == C.sizeof(want) and (ct in (C.c_float, C.c_double)) == (want in (C.c_float, C.c_double)), (name, prm, ct)


def test_distillation_forward_scopes_the_unrecorded_winograd_switch():
    """kernels.distillation_forward(flag) sets where no-grad / frozen convolutions may use the Winograd kernels and
    restores the previous setting, also when the body raises; defaults: teacher on, student's frozen trunk off."""
    from erd_amd import kernels as K
    assert K.WINO_TEACHER is True and K.WINO_FROZEN_TRUNK is False
    before = K.WINO_NOGRAD_FWD
    with K.distillation_forward(False):
        assert K.WINO_NOGRAD_FWD is False
        with K.distillation_forward(True):
            assert K.WINO_NOGRAD_FWD is True
        assert K.WINO_NOGRAD_FWD is False
    assert K.WINO_NOGRAD_FWD == before
    try:
        with K.distillation_forward():          # default argument: the frozen-trunk setting
            assert K.WINO_NOGRAD_FWD == K.WINO_FROZEN_TRUNK
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert K.WINO_NOGRAD_FWD == before


def test_resnet_init_cfg_pretrained_is_honoured(tmp_path, monkeypatch):
    """mmengine's Runner calls model.init_weights(): `init_cfg=dict(type='Pretrained', checkpoint=...)` loads the ImageNet
    backbone (torchvision keys map one to one), an unresolvable checkpoint raises instead of silently training on a
    frozen random trunk (configs/gfl_increment/*first_40_cats.py:45, resnet.py:305-420)."""
    import warnings
    import torch
    from erd_amd import MODELS
    torch.manual_seed(0)
    donor = MODELS.build(dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                              norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch"))
    sd = {k: (torch.randn_like(v) if v.dtype == torch.float32 else v.clone()) for k, v in donor.state_dict().items()}
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1000, 2048), torch.zeros(1000)      # torchvision's classifier: dropped
    hub = tmp_path / "ckpts"
    hub.mkdir()
    torch.save(sd, hub / "resnet50-0676ba61.pth")
    # decoys that sort BEFORE the canonical file: 'torchvision://resnet50' is one fixed file in the reference
    # (mmengine's torchvision-0.12 URL table), not "whatever resnet50*.pth comes first"
    torch.save({k: torch.zeros_like(v) for k, v in sd.items()}, hub / "resnet50-0000decoy.pth")
    cfg = dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
               norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, style="pytorch",
               init_cfg=dict(type="Pretrained", checkpoint="torchvision://resnet50"))
    monkeypatch.setenv("ERD_PRETRAINED_DIR", str(hub))
    net = MODELS.build(cfg)
    net.init_weights()
    own = net.state_dict()
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer4.2.conv3.weight", "layer3.5.bn2.bias"):
        assert torch.equal(own[k], sd[k]), k
    assert not net.conv1.weight.requires_grad and not net.layer1[0].conv1.weight.requires_grad      # still frozen
    assert net.layer2[0].conv1.weight.requires_grad
    # a plain file path works too; a missing file raises; the escape hatch warns
    MODELS.build({**cfg, "init_cfg": dict(type="Pretrained", checkpoint=str(hub / "resnet50-0676ba61.pth"))}).init_weights()
    monkeypatch.setenv("ERD_PRETRAINED_DIR", str(tmp_path / "nowhere"))
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "nohub"))
    with pytest.raises(FileNotFoundError):
        MODELS.build(cfg).init_weights()
    monkeypatch.setenv("ERD_ALLOW_RANDOM_BACKBONE", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        MODELS.build(cfg).init_weights()
    assert any("RANDOM" in str(x.message) for x in w)
    # without the canonical file a name glob still resolves -- with a warning naming what was picked
    monkeypatch.delenv("ERD_ALLOW_RANDOM_BACKBONE")
    only = tmp_path / "only"
    only.mkdir()
    torch.save(sd, only / "resnet50_custom.pth")
    monkeypatch.setenv("ERD_PRETRAINED_DIR", str(only))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        MODELS.build(cfg).init_weights()
    assert any("resnet50_custom.pth" in str(x.message) for x in w)
    # a checkpoint of the wrong depth is rejected
    monkeypatch.setenv("ERD_PRETRAINED_DIR", str(hub))
    with pytest.raises(RuntimeError):
        MODELS.build({**cfg, "depth": 101, "init_cfg": dict(type="Pretrained", checkpoint=str(hub / "resnet50-0676ba61.pth"))}).init_weights()


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md's binding table covers the whole C ABI: every function include/erd_hip.h declares is named there"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set(re.findall(r"\b(erd_[a-z0-9_]+)\s*\(", open(os.path.join(root, "include", "erd_hip.h")).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert not [n for n in sorted(names) if n not in doc]


def test_column_sum_replication_policy_fits_the_zero_arena():
    """kernels.colsum_copies: rows of the replicated column-sum accumulators of the input-gradient epilogues (a power of two between 8
    and 128, the accumulator at most COLSUM_WIDTH floats unless eight rows already exceed it), and the step's zero arena holds them for the
    deepest backbone of the configs (ResNet-101: 33 bottlenecks x three accumulators) with room for the row dots / d-gamma scratch."""
    import inspect
    from erd_amd import kernels as K
    assert K.COLSUM_WIDTH == 16384 and K.COLSUM_COPIES == 8
    for C, want in ((64, 128), (128, 128), (256, 64), (512, 32), (1024, 16), (2048, 8), (4096, 8), (72, 128), (1000, 16)):
        n = K.colsum_copies(C)
        assert n == want and n & (n - 1) == 0 and 8 <= n <= 128, (C, n)
        assert n * C <= max(K.COLSUM_WIDTH, 8 * C)
    planes = [64] * 3 + [128] * 4 + [256] * 23 + [512] * 3          # ResNet-101
    need = sum(4 * ((K.colsum_copies(p) * p) * 2 + K.colsum_copies(4 * p) * 4 * p) for p in planes)
    arena = inspect.signature(K.zero_arena_begin).parameters["nbytes"].default
    assert arena == 16 << 20 and need < arena // 2, (need, arena)
