#!/usr/bin/env python3
"""tower layer (conv3x3 256->256 on five levels + GroupNorm + ReLU) at the benched size: fused statistics against the three-pass form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K

sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
N, Cc = 4, 256
A = sum(h * w for h, w in sizes)
x = torch.randn(N, A, Cc, device="cuda")
w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.3


def timeit(fn, iters=20):
    best = 1e30
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


c = torch.empty(N, A, Cc, device="cuda")
print("conv only              %8.1f us" % timeit(lambda: K.conv_forward(K.level_views(x, sizes), w, K.level_views(c, sizes), 3, 1, 1)))
for f in (False, True, False, True):
    K.GN_FUSED = f
    print("layer, fused=%-5s     %8.1f us" % (f, timeit(lambda: K.conv3x3_gn_relu_forward(x, w, gamma, beta, sizes))))
print("gn_relu_forward alone  %8.1f us" % timeit(lambda: K.gn_relu_forward(c, gamma, beta, sizes)))
