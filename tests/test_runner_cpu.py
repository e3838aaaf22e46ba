"""CPU: the launcher's host logic -- LR schedule against torch's own schedulers, --cfg-options parsing, and the
checkpoint file format (reference keys, dense OIHW tensors, optional teacher copy)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_param_schedule_matches_torch_linear_and_multistep():
    from erd_amd import Config
    from erd_amd.runner import ParamSchedule
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py"))
    sch = ParamSchedule(cfg.param_scheduler)
    # schedule_1x.py:7-17: LinearLR(start .001, iters 0..500), MultiStepLR(milestones [8, 11], gamma .1)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    lin = torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.001, end_factor=1.0, total_iters=499)
    for it in range(520):
        assert sch.iter_factor(it) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-6), it
        opt.step(); lin.step()
    opt = torch.optim.SGD([p], lr=1.0)
    ms = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[8, 11], gamma=0.1)
    for ep in range(12):
        assert sch.epoch_factor(ep) == pytest.approx(opt.param_groups[0]["lr"], rel=1e-6), ep
        opt.step(); ms.step()


def test_cfg_options_parsing_and_merge():
    import train as T
    from erd_amd import Config
    o = T.parse_cfg_options(["train_dataloader.batch_size=4", "model.dist_loss_weight=0.5", "work_dir=foo",
                             "param_scheduler.1.milestones=[6,9]"])
    assert o == {"train_dataloader.batch_size": 4, "model.dist_loss_weight": 0.5, "work_dir": "foo",
                 "param_scheduler.1.milestones": [6, 9]}
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py"))
    cfg.merge_from_dict({"train_dataloader.batch_size": 4, "optim_wrapper.optimizer.lr": 0.5})
    assert cfg.train_dataloader.batch_size == 4 and cfg.optim_wrapper.optimizer.lr == 0.5


def test_checkpoint_format_roundtrip(tmp_path):
    from erd_amd import Config, MODELS
    from erd_amd.runner import load_checkpoint, model_state_dict, save_checkpoint
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py"))
    torch.manual_seed(0)
    m = MODELS.build(cfg.model)
    sd = model_state_dict(m)
    assert all(v.is_contiguous() for v in sd.values())                      # dense OIHW, whatever the live layout is
    assert sd["backbone.layer2.0.conv2.weight"].shape == (128, 128, 3, 3)
    assert sd["bbox_head.gfl_cls.weight"].shape == (40, 256, 3, 3) and "bbox_head.integral.project" in sd
    path = str(tmp_path / "epoch_1.pth")
    save_checkpoint(path, m, meta=dict(epoch=1))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"meta", "state_dict"} and ck["meta"]["epoch"] == 1
    torch.manual_seed(1)
    m2 = MODELS.build(cfg.model)
    assert not torch.equal(m2.state_dict()["bbox_head.gfl_reg.weight"], sd["bbox_head.gfl_reg.weight"])
    assert load_checkpoint(path, m2)["epoch"] == 1
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
