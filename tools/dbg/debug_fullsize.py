import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_gpu_fullsize as T
from erd_amd import functional as Fn
for aux in (True, False):
    Fn.WGRAD_ON_AUX_STREAM = aux
    for sk in (True, False, True, False):
        logs, model, t = T._run(sk, steps=2)
        print('aux', aux, 'sk', sk, [{k: round(v, 5) for k, v in l.items() if k in ('loss_dist_cls', 'loss_dist_bbox', 'loss_cls', 'loss')} for l in logs])
