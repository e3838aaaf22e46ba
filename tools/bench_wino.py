#!/usr/bin/env python3
"""Winograd F(2x2,3x3) vs the direct implicit-GEMM kernel on the 3x3 stride-1 shapes of the step (bs=4, fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erd_amd import kernels as K

N = int(os.environ.get("BS", "4"))


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def sizes_of(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out


CASES = [("head tower 5 levels", 256, 256, sizes_of(800, 1344)), ("head cls80", 256, 80, sizes_of(800, 1344)),
         ("fpn.out P3", 256, 256, [(100, 168)]), ("L1.conv2", 64, 64, [(200, 336)]), ("L2.conv2", 128, 128, [(100, 168)]),
         ("L3.conv2", 256, 256, [(50, 84)]), ("L4.conv2", 512, 512, [(25, 42)])]
print(f"{'layer':22s} {'GFLOP':>7s} | {'direct us':>9s} {'TF':>6s} | {'wino us':>8s} {'TF(alg)':>7s} | speedup")
ONLY = [o for o in os.environ.get("ONLY", "").split(",") if o]
for name, Cin, Cout, sizes in CASES:
    if ONLY and name not in ONLY:
        continue
    A = sum(h * w for h, w in sizes)
    x = torch.randn(N, A, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
    y = torch.empty(N, A, Cout, device="cuda")
    xs, ys = K.level_views(x, sizes), K.level_views(y, sizes)
    U = K.wino_weights(w)
    fl = 2.0 * N * A * Cout * Cin * 9
    K.WINOGRAD = False
    t_d = timeit(lambda: K.conv_forward(xs, w, ys, 3, 1, 1))
    K.WINOGRAD = True
    ref = y.clone()
    t_w = timeit(lambda: K.wino_conv3x3(xs, U, ys, Cout))
    err = float((y - ref).norm() / ref.norm())
    print(f"{name:22s} {fl/1e9:7.1f} | {t_d*1e3:9.1f} {fl/t_d/1e9:6.1f} | {t_w*1e3:8.1f} {fl/t_w/1e9:7.1f} | {t_d/t_w:5.2f}x  (rel diff {err:.1e})")
