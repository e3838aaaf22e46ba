"""GPU: the thin launcher end to end on a tiny workload -- config -> Runner -> epochs -> checkpoints -> resume.
(reference-format checkpoints, first-40 base training and the 40+40 incremental stage
chained through the checkpoint file exactly as the reference chains them, gfl_increment_erd.py:95-122.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_FIRST = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py")
CFG_INCRE = os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py")


def _cfg(path, work_dir, **over):
    from erd_amd import Config
    cfg = Config.fromfile(path)
    cfg.work_dir = str(work_dir)
    # (backbone.init_cfg=None as in the reference's own detector tests, test_single_stage.py:38-71: no ImageNet file here)
    cfg.merge_from_dict({"train_dataloader.batch_size": 2, "train_cfg.max_epochs": 2, "model.backbone.init_cfg": None,
                         "default_hooks.logger.interval": 1, **over})
    return cfg


def _data(num_classes, seed=0):
    from erd_amd.runner import SyntheticDetData
    return SyntheticDetData(2, num_classes, 2, image_hw=(123, 153), seed=seed)


def test_base_training_then_incremental_stage_then_resume(tmp_path):
    from erd_amd.runner import Runner
    torch.manual_seed(0)
    # stage 1: plain GFL on the first 40 classes, 2 epochs x 2 iterations
    r1 = Runner.from_cfg(_cfg(CFG_FIRST, tmp_path / "first40"), data=_data(40), log=lambda *_: None)
    h1 = r1.train()
    assert len(h1) == 4 and all(np.isfinite(r["loss"]) for r in h1)
    # config :112-116: SGD lr .01, auto_scale_lr on (x world*bs/16), LinearLR warm-up from x.001 by iteration
    assert h1[0]["lr"] == pytest.approx(0.01 * 2 / 16 * 0.001) and h1[1]["lr"] > h1[0]["lr"]
    base_ckpt = tmp_path / "first40" / "epoch_2.pth"
    # the file holds the weights AFTER the epoch's last (deferred) update: equal to the live model, and a checkpoint
    # written right after a step at a non-negligible learning rate differs from one written before that step's flush
    for k, v in torch.load(base_ckpt, map_location="cpu", weights_only=False)["state_dict"].items():
        assert torch.equal(v, r1.model.state_dict()[k].cpu()), k
    assert base_ckpt.is_file() and (tmp_path / "first40" / "last_checkpoint").read_text() == str(base_ckpt)
    ck = torch.load(base_ckpt, map_location="cpu", weights_only=False)
    assert ck["meta"]["epoch"] == 2 and ck["meta"]["iter"] == 4
    # the optimizer entry loads into a stock torch.optim.SGD over the same parameter list
    ps = [torch.nn.Parameter(p.detach().cpu().contiguous().clone()) for p in r1.model.parameters()]
    opt = torch.optim.SGD(ps, lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.load_state_dict(ck["optimizer"])
    assert len(opt.state) == sum(p.requires_grad for p in r1.model.parameters())

    # stage 2: the incremental detector reads stage 1's file as its teacher + warm start (ori_setting)
    over = {"model.ori_setting.ori_checkpoint_file": str(base_ckpt), "model.ori_setting.ori_config_file": CFG_FIRST}
    torch.manual_seed(5)              # the student's new-class rows are freshly initialised at build time
    r2 = Runner.from_cfg(_cfg(CFG_INCRE, tmp_path / "incre", **over), data=_data(40, seed=1), log=lambda *_: None)
    sd1 = ck["state_dict"]
    sd2 = r2.model.state_dict()
    assert torch.equal(sd2["ori_model.backbone.layer3.2.conv2.weight"].cpu(), sd1["backbone.layer3.2.conv2.weight"])
    assert torch.equal(sd2["bbox_head.gfl_cls.weight"][:40].cpu(), sd1["bbox_head.gfl_cls.weight"])
    h2 = r2.train()
    assert len(h2) == 4 and all(np.isfinite(r["loss"]) and "loss_dist_cls" in r for r in h2)
    final = {k: v.detach().cpu().clone() for k, v in r2.model.state_dict().items()}

    # resume: stop after epoch 1, resume from the work dir, finish epoch 2 -> same weights as the straight run
    torch.manual_seed(5)
    r3 = Runner.from_cfg(_cfg(CFG_INCRE, tmp_path / "incre_b", **over, **{"train_cfg.max_epochs": 1}),
                         data=_data(40, seed=1), log=lambda *_: None)
    r3.train()
    r4 = Runner.from_cfg(_cfg(CFG_INCRE, tmp_path / "incre_b", **over, resume=True), data=_data(40, seed=1),
                         log=lambda *_: None)
    assert r4.epoch == 1 and r4.trainer.iter == 2
    h4 = r4.train()
    assert [r["epoch"] for r in h4] == [2, 2]
    assert np.allclose([r["loss"] for r in h4], [r["loss"] for r in h2[2:]], rtol=1e-4)
    num = den = 0.0
    for k, v in r4.model.state_dict().items():
        if v.dtype == torch.float32 and not k.startswith("ori_model."):
            num += float((v.cpu() - final[k]).double().pow(2).sum()); den += float(final[k].double().pow(2).sum())
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5
    # teacher untouched by training, and a checkpoint written without it still loads
    assert torch.equal(r4.model.state_dict()["ori_model.bbox_head.gfl_cls.bias"].cpu(), sd1["bbox_head.gfl_cls.bias"])
    from erd_amd.runner import load_checkpoint, save_checkpoint
    slim = str(tmp_path / "slim.pth")
    save_checkpoint(slim, r4.model, with_teacher=False)
    assert not any(k.startswith("ori_model.") for k in torch.load(slim, weights_only=False)["state_dict"])
    load_checkpoint(slim, r2.model)
    assert torch.equal(r2.model.state_dict()["bbox_head.gfl_reg.weight"].cpu(), r4.model.state_dict()["bbox_head.gfl_reg.weight"].cpu())


@pytest.mark.parametrize("amp", [False, True])
def test_train_py_cli_smoke(tmp_path, amp):
    """--amp = the bf16 matrix-core mode (the reference's flag switches to AmpOptimWrapper, tools/train.py:82-92)"""
    import subprocess, sys
    teacher = tmp_path / "teacher.pth"
    from oracle import erd_oracle as O
    torch.save(dict(state_dict=O.procedural_state_dict(40, seed=0)), teacher)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train.py"), CFG_INCRE, "--work-dir", str(tmp_path / "w"),
           "--synthetic", "2", "--image-size", "123", "153", "--max-iters", "2", "--cfg-options",
           "train_dataloader.batch_size=2", f"model.ori_setting.ori_checkpoint_file={teacher}",
           f"model.ori_setting.ori_config_file={CFG_FIRST}", "default_hooks.logger.interval=1"]
    if amp:
        cmd.insert(3, "--amp")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Epoch(train) [1][2/2]" in out.stdout and "loss_dist_bbox" in out.stdout
