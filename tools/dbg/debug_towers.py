import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import bench
from e2e_util import build_erd, f7_state_dicts
from erd_amd import kernels as K
from erd_amd import functional as Fn
tsd, ssd = f7_state_dicts()
model = build_erd(tsd, ssd)
x, samples = bench.synthetic_gpu_batch(2, seed=10, device=torch.device('cuda', 0))
outs = {}
real_aux = Fn.aux_stream
for sk in (False, True):
    for conc in (False, True):
        K.STREAMK = sk
        Fn.aux_stream = real_aux if conc else (lambda dev: torch.cuda.current_stream(dev))
        with torch.no_grad():
            for rep in range(3):
                c, b, sizes = model._forward_cat(x)
                torch.cuda.synchronize()
                outs[(sk, conc, rep)] = (c.clone(), b.clone())
ref = outs[(False, False, 0)]
for k, v in outs.items():
    print(k, 'cls maxdiff %.3e bbox maxdiff %.3e' % (float((v[0]-ref[0]).abs().max()), float((v[1]-ref[1]).abs().max())))
print('---- teacher pass on side stream')
side = torch.cuda.Stream()
Fn.aux_stream = real_aux
touts = {}
for sk in (False, True):
    for use_side in (False, True):
        K.STREAMK = sk
        for rep in range(3):
            cur = torch.cuda.current_stream()
            if use_side:
                side.wait_stream(cur)
                with torch.cuda.stream(side), torch.no_grad():
                    t = model.teacher_pass(x)
                with torch.no_grad():
                    c, b, sizes = model._forward_cat(x)      # concurrent student forward
                cur.wait_stream(side)
            else:
                with torch.no_grad():
                    t = model.teacher_pass(x)
            torch.cuda.synchronize()
            touts[(sk, use_side, rep)] = (t.t_cls.clone(), t.t_bbox.clone(), t.ers['counts'].clone(), t.keep.clone())
ref = touts[(False, False, 0)]
for k, v in touts.items():
    print(k, 'cls %.3e bbox %.3e' % (float((v[0]-ref[0]).abs().max()), float((v[1]-ref[1]).abs().max())), v[2].flatten().tolist(), int(v[3].sum()))
