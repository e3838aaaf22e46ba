"""helpers shared by the GPU end-to-end tests / smoke / bench: build teacher+student through the registry
from the repo's config files, load procedural weights (oracle spec), make data samples."""
import os

import torch

import golden_inputs as G
from oracle import erd_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_FIRST = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py")
CFG_INCRE = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py")


CFG_FIRST70 = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r101_fpn_1x_coco_first_70_cats.py")
CFG_INCRE10 = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r101_fpn_1x_coco_first_70_incre_last_10_cats.py")


def f7_state_dicts(c_old=40, c_all=80, depth=50):
    tsd = O.procedural_state_dict(c_old, depth=depth, seed=0)
    ssd = O.student_state_from_teacher(tsd, c_all, seed=1)
    for k in sorted(ssd):
        if O.trainable(k) and ssd[k].dim() == 4:
            ssd[k] = ssd[k] + 0.02 * ssd[k].abs().mean() * G.randn(700 + len(k), *ssd[k].shape)
    return tsd, ssd


def build_erd(tsd, ssd, device="cuda", cfg_first=CFG_FIRST, cfg_incre=CFG_INCRE):
    import erd_amd
    from erd_amd import Config, MODELS
    tcfg = Config.fromfile(cfg_first)
    scfg = Config.fromfile(cfg_incre)
    scfg.model.latest_model_flag = False
    teacher = MODELS.build(tcfg.model)
    student = MODELS.build(scfg.model)
    teacher.load_state_dict(tsd, strict=True)
    student.load_state_dict(ssd, strict=True)
    student.attach_teacher(teacher, tsd["bbox_head.gfl_cls.bias"].shape[0])
    return student.to(device).train()


def make_samples(boxes, labels, metas, device="cuda"):
    from erd_amd import DetDataSample, InstanceData
    out = []
    for b, l, m in zip(boxes, labels, metas):
        ds = DetDataSample(metainfo=m)
        ds.gt_instances = InstanceData(bboxes=b.to(device), labels=l.to(device))
        out.append(ds)
    return out
