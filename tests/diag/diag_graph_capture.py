"""Bisect aid for whole-step stream capture (ERDTrainer(step_graph=True)): captures ONE sub-path of the step per mode
(A teacher on the origin stream, B* teacher / parts of it on a forked stream, C +losses, D +backward) so a HIP-runtime
crash in hipStreamEndCapture can be pinned to the stream topology that causes it.  Finding (round 2): joins into a
non-origin stream close a cycle in the runtime's parallel-capture-stream lists -> functional.CAPTURE_ORIGIN.
Run on the GPU box:  LD_PRELOAD=tools/dbg/segv_bt.so python tests/diag/diag_graph_capture.py D"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from types import SimpleNamespace
from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O
from erd_amd.engine import ERDTrainer
from erd_amd import kernels as K, functional as Fn, parse_losses
from erd_amd.structures import unpack_gt_instances
mode = sys.argv[1]
tsd, ssd = f7_state_dicts()
imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
x, metas = O.preprocess(imgs)
batch = (x.cuda(), make_samples(boxes, labels, metas))
model = build_erd(tsd, ssd)
tr = ERDTrainer(model, lr=0.02, batch_size_per_gpu=2, auto_scale_lr=False, warmup_iters=0)
dev = tr.device
FRESH = torch.cuda.Stream(device=dev)
gts, _, ms = unpack_gt_instances(batch[1])
counts = [int(g.bboxes.shape[0]) for g in gts]
N = 2
st = SimpleNamespace(x=torch.empty_like(batch[0]), gb=torch.zeros((N * 64, 4), device=dev), gl=torch.zeros((N * 64,), dtype=torch.long, device=dev),
                     goff=torch.zeros((N + 1,), dtype=torch.int32, device=dev), metas=[dict(pad_shape=tuple(m["pad_shape"])) for m in ms], names=None)
tr._fill_static(st, batch[0], gts, counts)

def body():
    cur = torch.cuda.current_stream(dev)
    if mode == "A":          # teacher only, on the capture stream
        with torch.no_grad():
            t = model.teacher_pass(st.x)
        return t.t_cls
    if mode == "B1":         # fork to the side stream, teacher only, join
        tr.side.wait_stream(cur)
        with torch.cuda.stream(tr.side), torch.no_grad():
            t = model.teacher_pass(st.x)
        cur.wait_stream(tr.side)
        return t.t_cls
    if mode in ("B6", "B7"):         # fork to the side stream, teacher only, no tower streams (no nested fork) / no trunk event
        if mode == "B6":
            Fn.TOWERS_ON_TWO_STREAMS = False
        else:
            os.environ["ERD_SHARE_TRUNK"] = "0"
        tr.side.wait_stream(cur)
        with torch.cuda.stream(tr.side), torch.no_grad():
            t = model.teacher_pass(st.x)
        cur.wait_stream(tr.side)
        return t.t_cls
    if mode in ("B12", "B13", "B14"):
        sd = tr.side
        with torch.no_grad():
            feats = model.ori_model.backbone(st.x)           # on the capture stream
            if mode != "B12":
                p_cat, sizes = model.ori_model.neck.forward_cat(feats)
        if mode == "B14":
            Fn.TOWERS_ON_TWO_STREAMS = False
        sd.wait_stream(cur)
        with torch.cuda.stream(sd), torch.no_grad():
            if mode == "B12":
                r = model.ori_model.neck.forward_cat(feats)[0]
            else:
                r = model.ori_model.bbox_head.forward_cat(p_cat, sizes)[0]
        cur.wait_stream(sd)
        return r
    if mode in ("B8", "B9", "B10", "B11"):
        sd = tr.side if mode != "B10" else FRESH
        sd.wait_stream(cur)
        with torch.cuda.stream(sd), torch.no_grad():
            if mode == "B9":
                r = st.x + 1.0
            elif mode == "B8":
                r = model.ori_model.backbone(st.x)[-1]
            elif mode == "B11":
                r = model.ori_model._forward_cat(st.x)[0]
            else:
                r = model.teacher_pass(st.x).t_cls
        cur.wait_stream(sd)
        return r
    if mode == "B2":         # fork, teacher + targets, join
        tr.side.wait_stream(cur)
        with torch.cuda.stream(tr.side), torch.no_grad():
            t = model.teacher_pass(st.x)
            t.targets = model.bbox_head._targets_packed(t.sizes, st.gb, st.gl, st.goff, 64, st.metas, dev)
        cur.wait_stream(tr.side)
        return t.t_cls
    if mode == "B3":         # no fork: teacher then student on the capture stream
        with torch.no_grad():
            t = model.teacher_pass(st.x)
        tr.flat.zero_grad(); K.zero_arena_begin(dev)
        with K.distillation_forward(K.WINO_FROZEN_TRUNK):
            s_cls, s_bbox, sizes = model._forward_cat(st.x, trunk=t.trunk)
        return s_cls
    if mode == "B4":         # student only
        tr.flat.zero_grad(); K.zero_arena_begin(dev)
        with K.distillation_forward(K.WINO_FROZEN_TRUNK):
            s_cls, s_bbox, sizes = model._forward_cat(st.x)
        return s_cls
    tr.side.wait_stream(cur)
    with torch.cuda.stream(tr.side), torch.no_grad():
        t = model.teacher_pass(st.x)
        t.targets = model.bbox_head._targets_packed(t.sizes, st.gb, st.gl, st.goff, 64, st.metas, dev)
    tr.flat.zero_grad(); K.zero_arena_begin(dev)
    if t.trunk is not None:
        cur.wait_event(t.trunk_event)
    with K.distillation_forward(K.WINO_FROZEN_TRUNK):
        s_cls, s_bbox, sizes = model._forward_cat(st.x, trunk=t.trunk)
    cur.wait_stream(tr.side)
    if mode == "B":
        return s_cls
    losses = model.bbox_head.loss_cat(t.t_cls, t.t_bbox, s_cls, s_bbox, sizes, None, t.ers, t.keep, model.ori_num_classes, model.dist_loss_weight, targets=t.targets)
    total, lv = parse_losses(losses)
    if mode == "C":
        return total
    total.backward()
    Fn.trail_join(dev)
    K.zero_arena_end()
    return total

cur = torch.cuda.current_stream(dev)
s = torch.cuda.Stream(device=dev); s.wait_stream(cur)
Fn.CAPTURE_ORIGIN = s.cuda_stream
with torch.cuda.stream(s):
    for _ in range(2): body()
s.synchronize()
print("warm-up ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = body()
print("capture ok", flush=True)
g.replay(); torch.cuda.synchronize()
print("replay ok", float(out.float().sum()), flush=True)
