import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import bench
from erd_amd.engine import ERDTrainer
dev = torch.device('cuda', 0)
model, cfg = bench.build_model(dev, 0)
tr = ERDTrainer(model, lr=0.01, batch_size_per_gpu=4, overlap_teacher=os.environ.get('OVERLAP', '1') == '1')
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev) for i in range(2)]
for i in range(8):
    log = tr.train_step(*batches[i % 2])
    torch.cuda.synchronize()
    print(i, {k: round(float(v.detach()), 5) for k, v in log.items()}, flush=True)
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if bad: print('  non-finite params:', bad[:5]); break
