#!/usr/bin/env python3
"""Where the host time of one training step goes: time inside the C-ABI calls (ctypes marshalling + hipLaunchKernel),
time inside the wrappers of erd_amd.kernels (descriptor building, checks), the rest (autograd, torch ops, modules).
usage: [ERD_COMPUTE=bf16] python tools/host_breakdown.py"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from erd_amd import _lib, kernels as K
from erd_amd.engine import ERDTrainer

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
for i in range(3):
    tr.train_step(*batches[i % 2])
tr.flush(); torch.cuda.synchronize()

acc = collections.defaultdict(lambda: [0, 0.0])
orig_call = _lib.call
def timed_call(name, *args):
    t = time.perf_counter()
    orig_call(name, *args)
    a = acc["C:" + name]; a[0] += 1; a[1] += time.perf_counter() - t
_lib.call = timed_call
K.call = timed_call
def wrap(name):
    f = getattr(K, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        q = acc["K:" + name]; q[0] += 1; q[1] += time.perf_counter() - t
        return r
    setattr(K, name, g)
for name in ["conv_forward", "conv_dgrad", "conv_wgrad_partials", "wgrad_reduce", "weight_transpose", "bn_fold", "bn_dgamma",
             "relu_bwd_colsum", "gn_relu_forward", "gn_relu_backward", "wino_conv3x3", "wino_weights", "zeros_f32", "to_bf16",
             "_weights_bf16"]:
    wrap(name)
# coarse sections of the step (host time spent inside each call)
def wrap_method(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        q = acc["S:" + tag]; q[0] += 1; q[1] += time.perf_counter() - t
        return r
    setattr(obj, name, g)
wrap_method(model, "teacher_pass", "teacher_pass (forward + ERS + NMS + targets)")
wrap_method(model, "_forward_cat", "student forward")
wrap_method(model.bbox_head, "loss_cat", "loss_cat")
wrap_method(tr, "_apply_pending", "sgd + shadow refresh")
wrap_method(tr.flat, "zero_grad", "zero_grad")
_bw = torch.Tensor.backward
def timed_backward(self, *a, **k):
    t = time.perf_counter()
    r = _bw(self, *a, **k)
    q = acc["S:backward (autograd thread, all Function.backward bodies)"]; q[0] += 1; q[1] += time.perf_counter() - t
    return r
torch.Tensor.backward = timed_backward
n = 8
t0 = time.perf_counter()
for i in range(n):
    tr.train_step(*batches[i % 2])
tr.flush()
host = time.perf_counter() - t0
torch.cuda.synchronize()
c_time = sum(v[1] for k, v in acc.items() if k.startswith("C:"))
c_calls = sum(v[0] for k, v in acc.items() if k.startswith("C:"))
k_time = sum(v[1] for k, v in acc.items() if k.startswith("K:") and k not in ("K:zeros_f32", "K:_weights_bf16", "K:to_bf16", "K:wino_weights"))
print(f"per step: host {1e3*host/n:.2f} ms; inside C-ABI calls {1e3*c_time/n:.2f} ms ({c_calls/n:.0f} calls, {1e6*c_time/c_calls:.1f} us each); "
      f"inside the listed kernels.py wrappers (incl. their C calls) {1e3*k_time/n:.2f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"  {k:66s} {v[0]/n:7.1f} calls/step {1e3*v[1]/n:7.3f} ms/step {1e6*v[1]/v[0]:7.1f} us/call")
