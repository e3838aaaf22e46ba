import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import golden_inputs as G
from e2e_util import build_erd, f7_state_dicts, make_samples
from oracle import erd_oracle as O
from erd_amd import parse_losses
tsd, ssd = f7_state_dicts()
model = build_erd(tsd, ssd)
imgs, boxes, labels = O.synthetic_batch(2, 123, 153, 40, seed=0)
x, metas = O.preprocess(imgs)
sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v) for k, v in ssd.items()}
with torch.no_grad():
    t_cls, t_bbox = O.gfl_forward(tsd, x)
s_cls, s_bbox = O.gfl_forward(sd, x)
for t in s_cls + s_bbox: t.retain_grad()
ref_losses, aux = O.erd_head_loss(t_cls, t_bbox, s_cls, s_bbox, boxes, labels, metas, 40, 80, 1.0, return_aux=True)
O.parse_losses(ref_losses).backward()
# mine
tc, tb, sizes = model.ori_model._forward_cat(x.cuda())
print('teacher cls err', float((tc.cpu()-O.flatten_levels(t_cls)).abs().max()), 'bbox', float((tb.cpu()-O.flatten_levels(t_bbox)).abs().max()))
sc, sb, sizes = model._forward_cat(x.cuda())
sc.retain_grad(); sb.retain_grad()
print('student cls err', float((sc.detach().cpu()-O.flatten_levels(s_cls)).abs().max()), 'bbox', float((sb.detach().cpu()-O.flatten_levels(s_bbox)).abs().max()))
ers = model.sel_pos_cat(tc, tb)
from erd_amd import kernels as K
anchors = model.bbox_head.prior_generator.grid_priors_cat(sizes, 'cuda')
keep,_ = K.distill_nms(tc, tb, anchors, ers['idx_bbox'], ers['counts'], 0.005)
losses = model.bbox_head.loss_cat(tc, tb, sc, sb, sizes, make_samples(boxes, labels, metas), ers, keep, 40, 1.0)
tot,_ = parse_losses(losses); tot.backward()
gc = torch.cat([t.grad.permute(0,2,3,1).reshape(2,-1,80) for t in s_cls],1); gb = torch.cat([t.grad.permute(0,2,3,1).reshape(2,-1,68) for t in s_bbox],1)
print('dcls err', float((sc.grad.cpu()-gc).abs().max()), float(gc.abs().max()), 'dbbox err', float((sb.grad.cpu()-gb).abs().max()), float(gb.abs().max()))
params = dict(model.named_parameters())
rows=[]
for k,v in sd.items():
    if not (O.trainable(k) and v.dtype==torch.float32): continue
    a,b = params[k].grad.cpu(), v.grad
    rows.append((float((a-b).abs().max())/(float(b.abs().max())+1e-20), k))
for e,k in rows:
    if e>2e-4: print('%.2e %s'%(e,k))
print('max', max(rows))
num = sum(float((params[k].grad.cpu().double()-v.grad.double()).pow(2).sum()) for k,v in sd.items() if O.trainable(k) and v.dtype==torch.float32)
den = sum(float(v.grad.double().pow(2).sum()) for k,v in sd.items() if O.trainable(k) and v.dtype==torch.float32)
print('GLOBAL L2 rel', (num/den)**0.5)
l2 = sorted(((float((params[k].grad.cpu()-v.grad).norm()/v.grad.norm()), k) for k,v in sd.items() if O.trainable(k) and v.dtype==torch.float32), reverse=True)
print('worst per-tensor L2 rel', l2[:5]); import numpy as np; print('median per-tensor L2 rel', np.median([a for a,_ in l2]))

