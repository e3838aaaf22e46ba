"""CPU: the oracle restatement (oracle/erd_oracle.py) against the fixtures produced by the
REAL reference source (oracle/gen_golden.py), plus the reference's own known-answer vectors."""
import numpy as np
import pytest
import torch

import golden_inputs as G
from oracle import erd_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach() if isinstance(a, torch.Tensor) else T(a)
    b = b.detach() if isinstance(b, torch.Tensor) else T(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a.float(), b.float(), rtol=rtol, atol=atol), float((a.float() - b.float()).abs().max())


def test_f1_leaf_losses(golden):
    g = golden("f1_leaf.npz")
    d = G.f1_inputs()
    p = d["qfl_pred"].clone().requires_grad_(True)
    rows = O.quality_focal_loss(p, d["qfl_label"], d["qfl_score"])
    close(rows, g["qfl_rows"])
    l = O.weight_reduce(rows, d["qfl_weight"], 37.0)
    l.backward()
    close(l, g["qfl_loss"]); close(p.grad, g["qfl_grad"])
    p = d["dfl_pred"].clone().requires_grad_(True)
    l = 0.25 * O.weight_reduce(O.distribution_focal_loss(p, d["dfl_label"]), d["dfl_weight"], 4.0)
    l.backward()
    close(l, g["dfl_loss"]); close(p.grad, g["dfl_grad"])
    p = d["kd_pred"].clone().requires_grad_(True)
    l = 0.25 * O.weight_reduce(O.kd_kl_div(p, d["kd_soft"], 10.0), d["kd_weight"], 4.0)
    l.backward()
    close(l, g["kd_loss"]); close(p.grad, g["kd_grad"], atol=1e-8)
    p = d["box_a"].clone().requires_grad_(True)
    l = O.giou_loss_module(p, d["box_b"], d["box_w"], 1.0)
    l.backward()
    close(l, g["giou_loss"]); close(p.grad, g["giou_grad"])
    close(O.bbox_overlaps(d["box_a"], d["box_b"], is_aligned=True), g["iou_aligned"])
    close(O.bbox_overlaps(d["box_a"], d["box_b"], mode="giou", is_aligned=True), g["giou_aligned"])
    close(O.bbox_overlaps(d["box_a"][:16], d["box_b"][:8]), g["iou_pair"])
    close(O.giou_loss_module(d["box_a"], d["box_b"], torch.zeros(128), 1.0), g["giou_zero_weight"])
    p = d["dfl_pred"].clone().requires_grad_(True)
    y = O.integral(p)
    (y * d["dist"][:96]).sum().backward()
    close(y, g["integral"]); close(p.grad, g["integral_grad"])
    close(O.distance2bbox(d["pts"], d["dist"]), g["distance2bbox"])
    close(O.bbox2distance(d["pts"], d["box_a"], 16, 0.1), g["bbox2distance"])
    close((d["l2_a"] - d["l2_b"]).pow(2).mean(), g["l2"])


def _sizes(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out


@pytest.mark.parametrize("H,W", [(800, 1344), (800, 1088), (1344, 800), (256, 256)])
def test_f2_anchors(golden, H, W):
    g = golden("f2_anchors.npz")
    key = f"{H}x{W}"
    sizes = _sizes(H, W)
    assert np.array_equal(np.array(sizes), g[key + "_sizes"])
    a = torch.cat(O.grid_anchors(sizes), 0)
    assert torch.equal(a[:4], T(g[key + "_first"])) and torch.equal(a[-4:], T(g[key + "_last"]))
    assert torch.equal(a[T(g[key + "_idx"])], T(g[key + "_sample"]))
    assert torch.equal(a.double().sum(0), T(g[key + "_sum"]))
    fl = O.valid_flags(sizes, (H - 37, W - 61))
    assert [int(f.sum()) for f in fl] == g[key + "_nvalid"].tolist()
    f = torch.cat(fl, 0)
    assert int((f.long() * torch.arange(f.numel())).sum()) == int(g[key + "_flagsum"])


def test_anchor_known_answers():
    # reference test_anchor_generator.py:636-641: 640x640 -> valid counts per level
    sizes = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    assert [int(f.sum()) for f in O.valid_flags(sizes, (640, 640))] == [6400, 1600, 400, 100, 25]
    # A = 22 400 anchors at 800x1344
    assert sum(h * w for h, w in _sizes(800, 1344)) == 22400
    a = O.grid_anchors([(2, 2)], strides=[16])[0]
    assert a.tolist() == [[-64., -64., 64., 64.], [-48., -64., 80., 64.],
                          [-64., -48., 64., 80.], [-48., -48., 80., 80.]]


def test_atss_known_answer():
    # reference tests/.../test_atss_assigner.py:12-36 : 4 priors x 2 gts -> gt_inds [1,0,0,0]
    priors = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]])
    gtb = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]])
    gtl = torch.LongTensor([2, 3])
    inds, labels = O.atss_assign(priors, [4], gtb, gtl, topk=9)
    assert inds.tolist() == [1, 0, 0, 0]
    assert labels.tolist() == [2, -1, -1, -1]
    # empty gt (:68-147 family): everything background
    inds, labels = O.atss_assign(priors, [4], torch.empty(0, 4), torch.empty(0, dtype=torch.long))
    assert inds.tolist() == [0, 0, 0, 0] and labels.tolist() == [-1] * 4


def test_f3_atss(golden):
    g = golden("f3_atss.npz")
    sizes = _sizes(800, 1344)
    anchors = torch.cat(O.grid_anchors(sizes), 0)
    nl = [h * w for h, w in sizes]
    for ci, case in enumerate(G.ATSS_CASES):
        gtb, gtl = G.atss_case(case)
        inds, labels = O.atss_assign(anchors, nl, gtb, gtl)
        pos = (inds > 0).nonzero().squeeze(1)
        assert np.array_equal(pos.numpy(), g[f"c{ci}_pos"])
        assert np.array_equal(inds[pos].numpy(), g[f"c{ci}_gt"])
        assert np.array_equal(labels[pos].numpy(), g[f"c{ci}_label"])


def test_f4_ers(golden):
    g = golden("f4_ers.npz")
    for i, seed in enumerate((400, 401, 402)):
        cls, bbox = G.ers_inputs(seed)
        ic, ib, tc, tb = O.ers_select_single(cls, bbox)
        assert np.array_equal(ic.numpy(), g[f"s{i}_cls_idx"])
        assert np.array_equal(ib.numpy(), g[f"s{i}_bbox_idx"])
        assert np.allclose([tc, tb], g[f"s{i}_thr"], rtol=1e-6)
        assert 0.01 < len(ic) / cls.shape[0] < 0.08 and 0.01 < len(ib) / cls.shape[0] < 0.08


def test_f5_nms_unpinned(golden):
    g = golden("f5_nms_unpinned.npz")
    for i, seed in enumerate((500, 510)):
        b, s, ids = G.nms_inputs(seed)
        for j, thr in enumerate((0.005, 0.6)):
            assert np.array_equal(O.nms_class_offset(b, s, ids, thr).numpy(), g[f"s{i}_t{j}_keep"])


@pytest.mark.parametrize("case", [0, 1])
def test_f8_predict_unpinned_nms(golden, case):
    """inference post-processing: the restatement reproduces the reference's GFLHead.predict_by_feat bit for bit
    (NMS itself = the mmcv restatement, unpinned)."""
    g = golden("f8_predict_unpinned_nms.npz")
    cls, bbox, metas, rescale = G.f8_inputs(case)
    res = O.predict_by_feat(cls, bbox, metas, rescale=rescale)
    for i, (b, s, l) in enumerate(res):
        assert np.array_equal(b.numpy(), g[f"c{case}_i{i}_bboxes"])
        assert np.array_equal(s.numpy(), g[f"c{case}_i{i}_scores"])
        assert np.array_equal(l.numpy(), g[f"c{case}_i{i}_labels"])
    assert len(res[0][0]) == (100 if case == 0 else 52)


def test_filter_scores_and_topk_known_answer():
    # misc.py:308-354 on a hand-made case: threshold is strict, order is score-descending, k caps the list
    scores = torch.tensor([[0.1, 0.9], [0.5, 0.05], [0.7, 0.3]])
    s, labels, keep = O.filter_scores_and_topk(scores, 0.05, 3)
    assert s.tolist() == pytest.approx([0.9, 0.7, 0.5]) and labels.tolist() == [1, 0, 0] and keep.tolist() == [0, 2, 1]
    s, labels, keep = O.filter_scores_and_topk(scores, 0.05, 100)
    assert len(s) == 5 and 0.05 not in s.tolist()


def test_weighted_loss_docstring_values():
    # losses/utils.py:80-96 docstring: mean 1.5, avg_factor=2 -> 3.0 (+eps)
    loss = torch.tensor([1.0, 2.0, 1.0, 2.0])   # l1(pred=[0,2,3], target=[1,0,1]) style values
    assert float(O.weight_reduce(torch.tensor([1.0, 2.0, 2.0]), None, None)) == pytest.approx(5 / 3)
    assert float(O.weight_reduce(torch.tensor([1.0, 2.0, 2.0]), torch.tensor([1.0, 0.0, 1.0]), 2)) == pytest.approx(1.5)
    del loss


def test_f6_head_losses_and_grads(golden):
    g = golden("f6_head.npz")
    sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas = G.f6_inputs()
    s_cls = [t.clone().requires_grad_(True) for t in s_cls]
    s_bbox = [t.clone().requires_grad_(True) for t in s_bbox]
    losses, aux = O.erd_head_loss(t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas, 40, 80, 1.0, return_aux=True)
    for k, vs in losses.items():
        close(torch.stack([v.detach() for v in vs]), g[k], rtol=1e-5, atol=1e-7)
    total = O.parse_losses(losses)
    assert float(total) == pytest.approx(float(g["total"]), rel=1e-5)
    total.backward()
    for l in range(5):
        close(s_cls[l].grad, g[f"g_cls{l}"], rtol=1e-4, atol=1e-8)
        close(s_bbox[l].grad, g[f"g_bbox{l}"], rtol=1e-4, atol=1e-8)
    for i in range(len(gtb)):
        assert np.array_equal(aux["ers_cls"][i].numpy(), g[f"ers_cls{i}"])
        assert np.array_equal(aux["ers_bbox"][i].numpy(), g[f"ers_bbox{i}"])


@pytest.mark.parametrize("fixture,c_old,depth,nparam", [("f7_tiny_e2e.npz", 40, 50, 32215193),
                                                        ("f9_tiny_e2e_r101_70_10.npz", 70, 101, 51207321)])
def test_f7_tiny_end_to_end(golden, fixture, c_old, depth, nparam):
    g = golden(fixture)
    tsd = O.procedural_state_dict(c_old, depth=depth, seed=0)
    ssd = O.student_state_from_teacher(tsd, 80, seed=1)
    for k in sorted(ssd):
        if O.trainable(k) and ssd[k].dim() == 4:
            ssd[k] = ssd[k] + 0.02 * ssd[k].abs().mean() * G.randn(700 + len(k), *ssd[k].shape)
    sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v)
          for k, v in ssd.items()}
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 80 - c_old, seed=0)
    x, metas = O.preprocess(imgs)
    t_cls, t_bbox = O.gfl_forward(tsd, x, depth)
    close(t_cls[0][0, :, ::4, ::4], g["teacher_cls0_sample"], rtol=1e-4, atol=1e-5)
    close(t_bbox[4], g["teacher_bbox4"], rtol=1e-4, atol=1e-5)
    losses = O.erd_step_loss(tsd, sd, x, boxes, labels, metas, c_old, 80, depth=depth)
    for k in ("loss_cls", "loss_bbox", "loss_dfl", "loss_dist_cls", "loss_dist_bbox"):
        close(torch.stack([v.detach() for v in losses[k]]), g[k], rtol=1e-4, atol=1e-7)
    total = O.parse_losses(losses)
    total.backward()
    names = [str(n) for n in g["grad_names"]]
    assert sorted(names) == sorted(k for k in sd if O.trainable(k))      # same trainable set
    assert sum(sd[k].numel() for k in names) == nparam, sum(sd[k].numel() for k in names)
    for i, k in enumerate(names):
        gr = sd[k].grad
        assert float(gr.double().norm()) == pytest.approx(float(g["grad_norms"][i]), rel=1e-3, abs=1e-9), k


@pytest.mark.parametrize("case", [0, 1])
def test_f10_head_losses_with_empty_ground_truth(golden, case):
    """one image / every image without boxes (reference fixture gen_f10): the `num_pos == 0` branches and the clamps of
    both normalisers"""
    g = golden("f10_head_empty_gt.npz")
    sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas = G.f10_inputs(case)
    s_cls = [t.clone().requires_grad_(True) for t in s_cls]
    s_bbox = [t.clone().requires_grad_(True) for t in s_bbox]
    losses = O.erd_head_loss(t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas, 40, 80, 1.0)
    for k, vs in losses.items():
        close(torch.stack([v.detach() for v in vs]), g[f"c{case}_{k}"], rtol=1e-5, atol=1e-7)
    total = O.parse_losses(losses)
    assert float(total) == pytest.approx(float(g[f"c{case}_total"]), rel=1e-5)
    total.backward()
    for l in range(5):
        close(s_cls[l].grad, g[f"c{case}_g_cls{l}"], rtol=1e-4, atol=1e-8)
        close(s_bbox[l].grad, g[f"c{case}_g_bbox{l}"], rtol=1e-4, atol=1e-8)
    if case == 1:
        assert all(float(v) == 0 for v in losses["loss_bbox"]) and all(float(v) == 0 for v in losses["loss_dfl"])


def test_bf16_modes_of_the_oracle_round_where_they_say():
    """`bf16_multiplicands` rounds the operands of the 1x1 / 3x3 convolutions only; `bf16_stored_maps` (round 5: the mode the unit-wise
    GPU test of BASELINE configs[2] compares against, tests/test_gpu_bf16_stagewise.py) additionally rounds every stored map and the
    gradient that flows back through it.  Outside the contexts the stage pieces (`resnet_stem` / `resnet_block` / `resnet_layer`,
    `head_tower_layer`) compose to exactly `gfl_forward` -- which the reference fixtures pin."""
    from e2e_util import f7_state_dicts
    tsd, ssd = f7_state_dicts()
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(3))
    sub = {k[len("backbone."):]: v for k, v in ssd.items() if k.startswith("backbone.")}
    rep = lambda t: torch.equal(t, t.to(torch.bfloat16).to(torch.float32))
    with torch.no_grad():
        ref = O.resnet_forward(ssd, x)
        h = O.resnet_stem(sub, x)
        for li in range(4):
            for b in range(O.RESNET_BLOCKS[50][li]):
                h = O.resnet_block(sub, h, li, b)
            assert torch.equal(h, ref[li])                                  # composition == the pinned forward, bit for bit
        assert not rep(ref[1])
        with O.bf16_multiplicands():
            m = O.resnet_forward(ssd, x)
        with O.bf16_stored_maps():
            s = O.resnet_forward(ssd, x)
            p = O.fpn_forward(ssd, s)
            c, r = O.gfl_head_forward(ssd, p)
        assert not rep(m[1]) and all(rep(t) for t in s) and all(rep(t) for t in p)     # maps are stored rounded ...
        assert not rep(c[0]) and not rep(r[0])                                          # ... head outputs stay fp32
        assert 1e-4 < float((m[3] - ref[3]).norm() / ref[3].norm()) < 3e-2
        assert 1e-4 < float((s[3] - ref[3]).norm() / ref[3].norm()) < 3e-2
    # the gradient through a stored map is rounded as well; outside the context `_store` is the identity
    t = torch.randn(4, 8, requires_grad=True)
    g = torch.randn(4, 8)
    with O.bf16_stored_maps():
        O._store(t * 1.0).backward(g)
    assert torch.equal(t.grad, g.to(torch.bfloat16).to(torch.float32))
    t.grad = None
    O._store(t * 1.0).backward(g)
    assert torch.equal(t.grad, g) and not O._BF16_STORED and not O._BF16_MULTIPLICANDS
