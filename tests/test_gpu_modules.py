"""GPU: the registered loss / coder / assigner MODULES invoked one at a time, the way a reference-side caller uses them
(`self.loss_cls(pred, (labels, score), weight=..., avg_factor=...)`, `self.bbox_coder.decode(...)`,
`self.assigner.assign(...)`, `self.integral(x)`): every value and gradient against what the REAL reference classes
produced on the same seeded inputs (fixtures F1 / F3, oracle/gen_golden.py) plus the reference's own known answers
(tests/test_models/test_task_modules/test_assigners/test_atss_assigner.py:12-36, test_losses/test_loss.py:18-27)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_inputs as G
from oracle import erd_oracle as O


@pytest.fixture(scope="module")
def reg():
    import erd_amd
    from erd_amd import MODELS, TASK_UTILS
    from erd_amd import _lib
    _lib.load()
    return MODELS, TASK_UTILS


def _close(a, b, rtol=2e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.allclose(a, b, rtol=rtol, atol=atol)


def test_quality_focal_loss_module(reg, golden):
    g, d = golden("f1_leaf.npz"), G.f1_inputs()
    qfl = reg[0].build(dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0))
    p = d["qfl_pred"].cuda().requires_grad_(True)
    tgt = (d["qfl_label"].cuda(), d["qfl_score"].cuda())
    loss = qfl(p, tgt, weight=d["qfl_weight"].cuda(), avg_factor=37.0)
    loss.backward()
    assert _close(loss, g["qfl_loss"]) and _close(p.grad, g["qfl_grad"], atol=1e-7)
    rows = qfl(p.detach(), tgt, reduction_override="none")
    assert _close(rows, g["qfl_rows"])
    # reduction rules of weight_reduce_loss (losses/utils.py:30-65)
    w = d["qfl_weight"].cuda()
    assert float(qfl(p.detach(), tgt, weight=w, reduction_override="sum")) == pytest.approx(float((g["qfl_rows"] * w.cpu().numpy()).sum()), rel=1e-5)
    assert float(qfl(p.detach(), tgt)) == pytest.approx(float(g["qfl_rows"].mean()), rel=1e-5)
    with pytest.raises(ValueError):
        qfl(p.detach(), tgt, avg_factor=3.0, reduction_override="sum")
    with pytest.raises(Exception):
        qfl(d["qfl_pred"], (d["qfl_label"], d["qfl_score"]))          # CPU tensors: no fallback


def test_distribution_focal_and_kd_modules(reg, golden):
    g, d = golden("f1_leaf.npz"), G.f1_inputs()
    dfl = reg[0].build(dict(type="DistributionFocalLoss", loss_weight=0.25))
    p = d["dfl_pred"].cuda().requires_grad_(True)
    loss = dfl(p, d["dfl_label"].cuda(), weight=d["dfl_weight"].cuda(), avg_factor=4.0)
    loss.backward()
    assert _close(loss, g["dfl_loss"]) and _close(p.grad, g["dfl_grad"], atol=1e-7)
    kd = reg[0].build(dict(type="KnowledgeDistillationKLDivLoss", loss_weight=0.25, T=10))
    p = d["kd_pred"].cuda().requires_grad_(True)
    loss = kd(p, d["kd_soft"].cuda(), weight=d["kd_weight"].cuda(), avg_factor=4.0)
    loss.backward()
    assert _close(loss, g["kd_loss"]) and _close(p.grad, g["kd_grad"], atol=1e-8)


def test_giou_loss_and_overlaps(reg, golden):
    g, d = golden("f1_leaf.npz"), G.f1_inputs()
    gi = reg[0].build(dict(type="GIoULoss", loss_weight=2.0))
    p = d["box_a"].cuda().requires_grad_(True)
    loss = gi(p, d["box_b"].cuda(), weight=d["box_w"].cuda(), avg_factor=1.0)
    loss.backward()
    assert _close(loss, g["giou_loss"]) and _close(p.grad, g["giou_grad"], rtol=1e-4, atol=1e-6)
    # zero weight -> 0 (test_loss.py:18-27 of the reference)
    assert float(gi(d["box_a"].cuda(), d["box_b"].cuda(), weight=torch.zeros(128).cuda(), avg_factor=1.0)) == float(g["giou_zero_weight"]) == 0.0
    ov = reg[1].build(dict(type="BboxOverlaps2D"))
    a, b = d["box_a"].cuda(), d["box_b"].cuda()
    assert _close(ov(a, b, "iou", True), g["iou_aligned"]) and _close(ov(a, b, "giou", True), g["giou_aligned"])
    assert _close(ov(a[:16], b[:8]), g["iou_pair"])
    assert ov(a[:0], b[:8]).shape == (0, 8)                              # empty input keeps its shape


def test_integral_and_bbox_coder(reg, golden):
    g, d = golden("f1_leaf.npz"), G.f1_inputs()
    from erd_amd.modules import Integral
    integ = Integral(16).cuda()
    p = d["dfl_pred"].cuda().requires_grad_(True)          # [4 * 96, 17] == [96, 68]
    y = integ(p.view(96, 68))
    assert y.shape == (96, 4)
    from erd_amd import leaf
    (y * d["dist"][:96].cuda()).sum().backward()          # (elementwise product / sum: torch glue of the TEST only)
    assert _close(y, g["integral"]) and _close(p.grad, g["integral_grad"], rtol=1e-4, atol=1e-5)
    coder = reg[1].build(dict(type="DistancePointBBoxCoder"))
    assert _close(coder.decode(d["pts"].cuda(), d["dist"].cuda()), g["distance2bbox"])
    assert _close(coder.encode(d["pts"].cuda(), d["box_a"].cuda(), 16, 0.1), g["bbox2distance"])
    # clip to the image, and the gradient of decode w.r.t. the distances (+-1 where the clamp is inactive)
    dist = d["dist"].cuda().requires_grad_(True)
    box = coder.decode(d["pts"].cuda(), dist, max_shape=(40, 60))
    ref = O.distance2bbox(d["pts"], d["dist"])
    ref[:, 0::2].clamp_(0, 60); ref[:, 1::2].clamp_(0, 40)
    assert _close(box, ref.numpy())
    box.sum().backward()
    raw = O.distance2bbox(d["pts"], d["dist"])
    inside = torch.stack([(raw[:, 0] >= 0) & (raw[:, 0] <= 60), (raw[:, 1] >= 0) & (raw[:, 1] <= 40),
                          (raw[:, 2] >= 0) & (raw[:, 2] <= 60), (raw[:, 3] >= 0) & (raw[:, 3] <= 40)], 1).float()
    assert torch.equal(dist.grad.cpu(), inside * torch.tensor([-1., -1., 1., 1.]))


def test_atss_assigner_module_known_answer_and_fixture(reg, golden):
    from erd_amd import InstanceData
    asg = reg[1].build(dict(type="ATSSAssigner", topk=9))
    # the reference's own known answer (test_atss_assigner.py:12-36): 4 priors x 2 gts -> gt_inds [1, 0, 0, 0]
    priors = torch.tensor([[0., 0., 10., 10.], [10., 10., 20., 20.], [5., 5., 15., 15.], [32., 32., 38., 42.]]).cuda()
    gts = InstanceData(bboxes=torch.tensor([[0., 0., 10., 9.], [0., 10., 10., 19.]]).cuda(), labels=torch.tensor([2, 3]).cuda())
    res = asg.assign(InstanceData(priors=priors), [4], gts)
    assert res.gt_inds.cpu().tolist() == [1, 0, 0, 0] and res.labels.cpu().tolist() == [2, -1, -1, -1] and res.num_gts == 2
    assert float(res.max_overlaps[0]) == pytest.approx(0.9) and float(res.max_overlaps[1]) == -100000000.0
    # no ground truth / no priors (test_atss_assigner.py:68-147)
    empty = InstanceData(bboxes=torch.zeros((0, 4)).cuda(), labels=torch.zeros((0,), dtype=torch.long).cuda())
    res0 = asg.assign(InstanceData(priors=priors), [4], empty)
    assert res0.gt_inds.cpu().tolist() == [0, 0, 0, 0] and res0.labels.cpu().tolist() == [-1] * 4 and res0.num_gts == 0
    assert asg.assign(InstanceData(priors=priors[:0]), [0], gts).gt_inds.numel() == 0
    # F3: the real reference assigner on the 800x1344 anchor grid, 4 seeded gt sets
    from erd_amd import kernels as K
    g = golden("f3_atss.npz")
    sizes, h, w = [], 100, 168
    for _ in range(5):
        sizes.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    anchors = K.grid_anchors(sizes, O.STRIDES, "cuda")
    sampler = reg[1].build(dict(type="PseudoSampler"))
    for ci, case in enumerate(G.ATSS_CASES):
        boxes, labels = G.atss_case(case)
        gi = InstanceData(bboxes=boxes.cuda(), labels=labels.cuda())
        pi = InstanceData(priors=anchors)
        res = asg.assign(pi, [a * b for a, b in sizes], gi)
        pos = (res.gt_inds > 0).nonzero().squeeze(1).cpu()
        assert np.array_equal(pos.numpy(), g[f"c{ci}_pos"])
        assert np.array_equal(res.gt_inds.cpu()[pos].numpy(), g[f"c{ci}_gt"])
        assert np.array_equal(res.labels.cpu()[pos].numpy(), g[f"c{ci}_label"])
        iou = O.bbox_overlaps(torch.cat(O.grid_anchors(sizes), 0)[pos], boxes[res.gt_inds.cpu()[pos] - 1], is_aligned=True)
        assert torch.allclose(res.max_overlaps.cpu()[pos], iou, rtol=1e-6, atol=1e-7)
        sr = sampler.sample(res, pi, gi)
        assert torch.equal(sr.pos_inds.cpu(), pos) and sr.neg_inds.numel() == anchors.shape[0] - pos.numel()
        assert torch.equal(sr.pos_gt_bboxes.cpu(), boxes[res.gt_inds.cpu()[pos] - 1]) and sr.avg_factor == max(pos.numel(), 1)
