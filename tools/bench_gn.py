#!/usr/bin/env python3
"""time the GroupNorm(32)+ReLU forward / backward kernels on the head's [4, 22400, 256] maps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erd_amd import kernels as K

sizes, h, w = [], 100, 168
for _ in range(5):
    sizes.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
A = sum(a * b for a, b in sizes)
c = torch.randn(4, A, 256, device="cuda")
dy = torch.randn(4, A, 256, device="cuda")
g = torch.rand(256, device="cuda") + 0.5
b = torch.randn(256, device="cuda") * 0.1


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


y, mr = K.gn_relu_forward(c, g, b, sizes, 32, 1e-5)
print("fwd  %.1f us (min traffic %.0f MB)" % (timeit(lambda: K.gn_relu_forward(c, g, b, sizes, 32, 1e-5)), 3 * c.numel() * 4 / 1e6))
print("bwd  %.1f us (min traffic %.0f MB)" % (timeit(lambda: K.gn_relu_backward(c, dy, g, b, mr, sizes, 32)), 5 * c.numel() * 4 / 1e6))
