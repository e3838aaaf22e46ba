#!/usr/bin/env python3
"""us per launch of the head-tower and L2.conv2 Winograd launches with ERD_WINO_P from the environment (probe libraries via ERD_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K
def sizes_of(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
N = 4
out = []
for name, Cin, Cout, sizes in [("head tower", 256, 256, sizes_of(800, 1344)), ("fpn.out P3", 256, 256, [(100, 168)]), ("L2.conv2", 128, 128, [(100, 168)])]:
    A = sum(h * w for h, w in sizes)
    x = torch.randn(N, A, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
    y = torch.empty(N, A, Cout, device="cuda")
    xs, ys = K.level_views(x, sizes), K.level_views(y, sizes)
    U = K.wino_weights(w, x3=True)
    t = min(timeit(lambda: K.wino_conv3x3(xs, U, ys, Cout)) for _ in range(2))
    out.append(f"{name} {t:7.1f}")
print(f"P={os.environ.get('ERD_WINO_P','-')} lib={os.path.basename(os.environ.get('ERD_HIP_LIB','shipped'))}: " + " | ".join(out))
