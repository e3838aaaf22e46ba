#!/usr/bin/env python3
"""which ATen ops (copies, fills, adds -- everything that is not a liberd_hip launch) one training step issues, with the Python
line that called them: python tools/dbg/aten_ops.py [f32x3|bf16]"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer
from torch.utils._python_dispatch import TorchDispatchMode

mode = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
K.set_compute(mode)
dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev, 0)
tr = ERDTrainer(model, lr=0.01, batch_size_per_gpu=4)
batches = [bench.synthetic_gpu_batch(4, s, dev, cfg) for s in (0, 1)]
for i in range(3):
    tr.train_step(*batches[i % 2], next_batch=batches[(i + 1) % 2])
torch.cuda.synchronize()

SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.as_strided", "aten.permute", "aten.slice", "aten.select", "aten.t.",
        "aten.expand", "aten.alias", "aten.reshape", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.unbind", "aten.split",
        "aten.empty", "aten.new_empty", "aten.is_", "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "aten.record_stream")
count = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if fr.filename.startswith(ROOT) and "dbg/aten_ops" not in fr.filename:
                    site = "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno)
                    break
            numel = next((a.numel() for a in args if isinstance(a, torch.Tensor)), 0)
            count[(name, site, numel)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    tr.train_step(*batches[1], next_batch=batches[0])
torch.cuda.synchronize()
tot = 0
for (name, site, numel), n in sorted(count.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%4d  %-34s %-38s numel %d" % (n, name, site, numel))
    tot += n
print("total", tot)
