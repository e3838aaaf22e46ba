import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import golden_inputs as G
from oracle import erd_oracle as O
import erd_amd
from erd_amd import MODELS
sd_full = O.procedural_state_dict(40, seed=0)
sd = {k[len('backbone.'):]: v for k, v in sd_full.items() if k.startswith('backbone.')}
m = MODELS.build(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0,1,2,3), frozen_stages=1, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'))
m.load_state_dict(sd, strict=True); m = m.cuda().train()
x = G.randn(1, 2, 3, 128, 160)
ref_sd = {('backbone.'+k): (v.clone().requires_grad_(True) if O.trainable('backbone.'+k) and v.dtype==torch.float32 else v) for k, v in sd.items()}
outs_ref = O.resnet_forward(ref_sd, x)
dys = [G.randn(10+i, *o.shape) for i, o in enumerate(outs_ref)]
which = [int(a) for a in sys.argv[1:]] or [1,2,3]
sum((outs_ref[i]*dys[i]).sum() for i in which).backward()
outs = m(x.cuda())
for i in range(4):
    print('fwd', i, float((outs[i].cpu()-outs_ref[i]).abs().max()/outs_ref[i].abs().max()))
sum((outs[i]*dys[i].cuda()).sum() for i in which).backward()
p = dict(m.named_parameters())
bad = 0
for k, v in ref_sd.items():
    if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None:
        a = p[k[len('backbone.'):]].grad.cpu()
        e = float((a - v.grad).abs().max()/(v.grad.abs().max()+1e-20))
        if e > 2e-4:
            bad += 1
            if bad < 40: print('%.2e %s' % (e, k))
print('bad', bad)
