"""TEST INFRASTRUCTURE -- writes ``tests/golden/*.npz`` by running the REAL reference
source (``/root/reference``, via ``oracle/ref_stub.py``) on the seeded inputs of
``tests/golden_inputs.py``.  Run in the build container only:

    python oracle/gen_golden.py

The fixtures are data (inputs are regenerated from seeds; expected outputs are
stored).  F5 (NMS) comes from the restatement of mmcv's published algorithm and
is flagged UNPINNED vs. mmcv==2.0.0 (mmcv is not in the image)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import erd_oracle as O  # noqa: E402  (only for the procedural-weight / input spec)
from oracle import ref_stub  # noqa: E402
import golden_inputs as G  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def npy(t):
    return t.detach().cpu().numpy()


def gen_f1(ref):
    d = G.f1_inputs()
    out = {}
    # QFL (gfocal_loss.py:168-249)
    qfl = ref.QualityFocalLoss(use_sigmoid=True, beta=2.0, loss_weight=1.0)
    p = d["qfl_pred"].clone().requires_grad_(True)
    l = qfl(p, (d["qfl_label"], d["qfl_score"]), weight=d["qfl_weight"], avg_factor=37.0)
    l.backward()
    out["qfl_loss"], out["qfl_grad"] = npy(l), npy(p.grad)
    out["qfl_rows"] = npy(ref.quality_focal_loss(d["qfl_pred"], (d["qfl_label"], d["qfl_score"]),
                                                 beta=2.0, reduction="none"))
    # DFL (gfocal_loss.py:252-295)
    dfl = ref.DistributionFocalLoss(loss_weight=0.25)
    p = d["dfl_pred"].clone().requires_grad_(True)
    l = dfl(p, d["dfl_label"], weight=d["dfl_weight"], avg_factor=4.0)
    l.backward()
    out["dfl_loss"], out["dfl_grad"] = npy(l), npy(p.grad)
    # KD-KL (kd_loss.py:40-95)
    kd = ref.KnowledgeDistillationKLDivLoss(loss_weight=0.25, T=10)
    p = d["kd_pred"].clone().requires_grad_(True)
    l = kd(p, d["kd_soft"], weight=d["kd_weight"], avg_factor=4.0)
    l.backward()
    out["kd_loss"], out["kd_grad"] = npy(l), npy(p.grad)
    # GIoU (iou_loss.py:463-528) + overlaps
    gi = ref.GIoULoss(loss_weight=2.0)
    p = d["box_a"].clone().requires_grad_(True)
    l = gi(p, d["box_b"], weight=d["box_w"], avg_factor=1.0)
    l.backward()
    out["giou_loss"], out["giou_grad"] = npy(l), npy(p.grad)
    out["iou_aligned"] = npy(ref.bbox_overlaps(d["box_a"], d["box_b"], is_aligned=True))
    out["giou_aligned"] = npy(ref.bbox_overlaps(d["box_a"], d["box_b"], mode="giou", is_aligned=True, eps=1e-6))
    out["iou_pair"] = npy(ref.bbox_overlaps(d["box_a"][:16], d["box_b"][:8]))
    out["giou_zero_weight"] = npy(gi(d["box_a"], d["box_b"], weight=torch.zeros(128), avg_factor=1.0))
    # Integral / coder
    integ = ref.modules["mmdet.models.dense_heads.gfl_head_increment_erd"].Integral(16)
    p = d["dfl_pred"].clone().requires_grad_(True)
    y = integ(p)
    (y * d["dist"][:96]).sum().backward()
    out["integral"], out["integral_grad"] = npy(y), npy(p.grad)
    out["distance2bbox"] = npy(ref.distance2bbox(d["pts"], d["dist"]))
    out["bbox2distance"] = npy(ref.bbox2distance(d["pts"], d["box_a"], 16, 0.1))
    out["l2"] = npy(ref.GFLHeadIncrementERD.l2_loss(d["l2_a"], d["l2_b"]))
    np.savez_compressed(os.path.join(OUT, "f1_leaf.npz"), **out)


def gen_f2(ref):
    ag = ref.AnchorGenerator(ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                             strides=[8, 16, 32, 64, 128])
    out = {}
    for (H, W) in [(800, 1344), (800, 1088), (1344, 800), (256, 256)]:
        sizes = [(int(np.ceil(H / s)), int(np.ceil(W / s))) for s in (8, 16, 32, 64, 128)]
        # featmap sizes follow the conv arithmetic (3x3 s2 p1): ceil division chain
        sizes = []
        h, w = H // 8, W // 8
        for _ in range(5):
            sizes.append((h, w))
            h, w = (h + 1) // 2, (w + 1) // 2
        anchors = torch.cat(ag.grid_priors(sizes, device="cpu"), 0)
        key = f"{H}x{W}"
        out[key + "_sizes"] = np.array(sizes)
        out[key + "_first"] = npy(anchors[:4])
        out[key + "_last"] = npy(anchors[-4:])
        idx = torch.linspace(0, anchors.shape[0] - 1, 64).long()
        out[key + "_idx"] = npy(idx)
        out[key + "_sample"] = npy(anchors[idx])
        out[key + "_sum"] = npy(anchors.double().sum(0))
        flags = torch.cat(ag.valid_flags(sizes, (H - 37, W - 61), device="cpu"), 0)
        out[key + "_nvalid"] = np.array([int(f.sum()) for f in ag.valid_flags(sizes, (H - 37, W - 61), device="cpu")])
        out[key + "_flagsum"] = np.array(int((flags.long() * torch.arange(flags.numel())).sum()))
    np.savez_compressed(os.path.join(OUT, "f2_anchors.npz"), **out)


def _sizes_800x1344():
    return [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]


def gen_f3(ref):
    ag = ref.AnchorGenerator(ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                             strides=[8, 16, 32, 64, 128])
    sizes = _sizes_800x1344()
    per_level = ag.grid_priors(sizes, device="cpu")
    anchors = torch.cat(per_level, 0)
    nl = [a.shape[0] for a in per_level]
    assigner = ref.ATSSAssigner(topk=9)
    out = {}
    for ci, case in enumerate(G.ATSS_CASES):
        gtb, gtl = G.atss_case(case)
        res = assigner.assign(ref.InstanceData(priors=anchors), nl,
                              ref.InstanceData(bboxes=gtb, labels=gtl))
        pos = (res.gt_inds > 0).nonzero().squeeze(1)
        out[f"c{ci}_pos"] = npy(pos)
        out[f"c{ci}_gt"] = npy(res.gt_inds[pos])
        out[f"c{ci}_label"] = npy(res.labels[pos])
    np.savez_compressed(os.path.join(OUT, "f3_atss.npz"), **out)


def gen_f4(ref):
    det = ref.GFLIncrementERD.__new__(ref.GFLIncrementERD)
    out = {}
    for i, seed in enumerate((400, 401, 402)):
        cls, bbox = G.ers_inputs(seed)
        ic, _, ib, _ = ref.GFLIncrementERD.sel_pos_single(det, cls, bbox)
        m_c = cls.sigmoid().max(-1)[0]
        m_b = bbox.max(-1)[0]
        thr_c = m_c.mean() + 2 * m_c.std()
        thr_b = m_b.mean() + 2 * m_b.std()
        out[f"s{i}_cls_idx"] = npy(ic)
        out[f"s{i}_bbox_idx"] = npy(ib)
        out[f"s{i}_thr"] = np.array([float(thr_c), float(thr_b)], dtype=np.float32)
        out[f"s{i}_margin"] = np.array([float((m_c - thr_c).abs().min()), float((m_b - thr_b).abs().min())])
    np.savez_compressed(os.path.join(OUT, "f4_ers.npz"), **out)


def gen_f5(ref):
    out = {}
    for i, seed in enumerate((500, 510)):
        b, s, ids = G.nms_inputs(seed)
        for j, thr in enumerate((0.005, 0.6)):
            _, keep = ref_stub.batched_nms(b, s, ids, dict(iou_threshold=thr))
            out[f"s{i}_t{j}_keep"] = npy(keep)
    np.savez_compressed(os.path.join(OUT, "f5_nms_unpinned.npz"), **out)


def _samples(ref, gtb, gtl, metas):
    samples = []
    for i in range(len(gtb)):
        ds = ref.DetDataSample(metainfo=metas[i])
        ds.gt_instances = ref.InstanceData(bboxes=gtb[i], labels=gtl[i])
        samples.append(ds)
    return samples


def gen_f6(ref):
    sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas = G.f6_inputs()
    teacher, student = ref_stub.build_reference_erd()
    ref_stub.attach_teacher(student, teacher, 40)
    s_cls = [t.clone().requires_grad_(True) for t in s_cls]
    s_bbox = [t.clone().requires_grad_(True) for t in s_bbox]
    ic, sc, ib, sb = student.sel_pos(t_cls, t_bbox)
    losses = student.bbox_head.loss((t_cls, t_bbox), (s_cls, s_bbox), _samples(ref, gtb, gtl, metas),
                                    ic, sc, ib, sb, 40, 1, student)
    total = O.parse_losses(losses)
    total.backward()
    out = {k: np.array([float(v) for v in vs], dtype=np.float32) for k, vs in losses.items()}
    out["total"] = np.array(float(total))
    for l in range(5):
        out[f"g_cls{l}"] = npy(s_cls[l].grad)
        out[f"g_bbox{l}"] = npy(s_bbox[l].grad)
    for i in range(len(ic)):
        out[f"ers_cls{i}"] = npy(ic[i])
        out[f"ers_bbox{i}"] = npy(ib[i])
    np.savez_compressed(os.path.join(OUT, "f6_head.npz"), **out)


def _tiny_e2e(ref, fname, c_old, c_all, depth):
    """tiny end-to-end: real ResNet/FPN/GFL teacher+student at 128x160, procedural weights."""
    teacher, student = ref_stub.build_reference_erd(c_old, c_all, depth)
    tsd = O.procedural_state_dict(c_old, depth=depth, seed=0)
    ssd = O.student_state_from_teacher(tsd, c_all, seed=1)
    for k in sorted(ssd):
        if O.trainable(k) and ssd[k].dim() == 4:       # move the student off the teacher
            ssd[k] = ssd[k] + 0.02 * ssd[k].abs().mean() * G.randn(700 + len(k), *ssd[k].shape)
    teacher.load_state_dict(tsd, strict=True)
    student.load_state_dict(ssd, strict=True)
    ref_stub.attach_teacher(student, teacher, c_old)
    student.train()
    imgs, boxes, labels = O.synthetic_batch(2, 123, 153, c_all - c_old, seed=0)
    x, metas = O.preprocess(imgs)
    losses = student.loss(x, _samples(ref, boxes, labels, metas))
    total = O.parse_losses(losses)
    total.backward()
    out = {k: np.array([float(v) for v in vs], dtype=np.float32) for k, vs in losses.items()}
    out["total"] = np.array(float(total))
    names, norms, samp = [], [], []
    for k, p in student.named_parameters():
        if k.startswith("ori_model.") or p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        flat = p.grad.reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, min(8, flat.numel())).long()
        s = torch.zeros(8)
        s[:idx.numel()] = flat[idx]
        samp.append(npy(s))
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["grad_samples"] = np.stack(samp)
    t_cls, t_bbox = teacher(x)
    out["teacher_cls0_sample"] = npy(t_cls[0][0, :, ::4, ::4])
    out["teacher_bbox4"] = npy(t_bbox[4])
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_f7(ref):
    _tiny_e2e(ref, "f7_tiny_e2e.npz", 40, 80, 50)


def gen_f9(ref):
    """BASELINE.json configs[3]: ResNet-101, 70 old + 10 new classes"""
    _tiny_e2e(ref, "f9_tiny_e2e_r101_70_10.npz", 70, 80, 101)


def gen_f8(ref):
    """inference post-processing through the reference's GFLHead.predict_by_feat (mmcv NMS = the stub restatement,
    UNPINNED).  Asserts the fixture's decisions have margin: no score within 1e-6 of score_thr or of the top-k cut."""
    _, student = ref_stub.build_reference_erd()
    teacher40, _ = ref_stub.build_reference_erd()
    out = {}
    for case in (0, 1):
        cls, bbox, metas, rescale = G.f8_inputs(case)
        head = student.bbox_head if cls[0].shape[1] == 80 else teacher40.bbox_head
        with torch.no_grad():
            res = head.predict_by_feat(cls, bbox, batch_img_metas=metas, rescale=rescale)
        for i, r in enumerate(res):
            out[f"c{case}_i{i}_bboxes"] = npy(r.bboxes)
            out[f"c{case}_i{i}_scores"] = npy(r.scores)
            out[f"c{case}_i{i}_labels"] = npy(r.labels)
            for cm in cls:
                s = cm[i].permute(1, 2, 0).reshape(-1).sigmoid()
                v = s[s > 0.05].sort(descending=True).values
                if v.numel() > 1000:                 # the top-k cut decides
                    assert float(v[999] - v[1000]) > 1e-6
                else:                                # the score threshold decides
                    assert float((s - 0.05).abs().min()) > 1e-6
    np.savez_compressed(os.path.join(OUT, "f8_predict_unpinned_nms.npz"), **out)


def gen_f10(ref):
    """head-level losses + gradients with EMPTY ground truth (one image / every image) through the reference's
    GFLHeadIncrementERD.loss: pins the `num_pos == 0` branches (gfl_head_increment_erd.py:293-297), the clamp of both
    normalisers to >= 1 (:379-398) and the assigner's no-GT path (atss_assigner.py:120-134)."""
    out = {}
    for case in (0, 1):
        sizes, t_cls, t_bbox, s_cls, s_bbox, gtb, gtl, metas = G.f10_inputs(case)
        teacher, student = ref_stub.build_reference_erd()
        ref_stub.attach_teacher(student, teacher, 40)
        s_cls = [t.clone().requires_grad_(True) for t in s_cls]
        s_bbox = [t.clone().requires_grad_(True) for t in s_bbox]
        ic, sc, ib, sb = student.sel_pos(t_cls, t_bbox)
        losses = student.bbox_head.loss((t_cls, t_bbox), (s_cls, s_bbox), _samples(ref, gtb, gtl, metas),
                                        ic, sc, ib, sb, 40, 1, student)
        total = O.parse_losses(losses)
        total.backward()
        for k, vs in losses.items():
            out[f"c{case}_{k}"] = np.array([float(v) for v in vs], dtype=np.float32)
        out[f"c{case}_total"] = np.array(float(total))
        for l in range(5):
            out[f"c{case}_g_cls{l}"] = npy(s_cls[l].grad)
            out[f"c{case}_g_bbox{l}"] = npy(s_bbox[l].grad)
    np.savez_compressed(os.path.join(OUT, "f10_head_empty_gt.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = ref_stub.load_reference()
    only = set(sys.argv[1:])                 # e.g. `python oracle/gen_golden.py gen_f10`
    for fn in (gen_f1, gen_f2, gen_f3, gen_f4, gen_f5, gen_f6, gen_f7, gen_f8, gen_f9, gen_f10):
        if only and fn.__name__ not in only:
            continue
        fn(ref)
        print("wrote", fn.__name__)


if __name__ == "__main__":
    main()
