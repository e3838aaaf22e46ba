#!/usr/bin/env python3
"""microseconds per launch of the Winograd shapes of the step (three-limb form), sixteen-wave against eight-wave kernel:
   python tools/bench_wino_wide.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erd_amd import kernels as K, _lib
from erd_amd.kernels import level_views

lib = _lib.load()
N = int(os.environ.get("BS", "4"))


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


CASES = [("head tower 5 levels 256->256", [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)], 256, 256),
         ("fpn.out P3 256->256", [(100, 168)], 256, 256), ("L2.conv2 128->128", [(100, 168)], 128, 128),
         ("L3.conv2 256->256", [(50, 84)], 256, 256), ("L4.conv2 512->512", [(25, 42)], 512, 512),
         ("head cls 256->80", [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)], 256, 80)]
print("%-32s %10s %10s" % ("launch", "wide us", "8-wave us"))
for name, sizes, Cin, Cout in CASES:
    A = sum(h * w for h, w in sizes)
    x = torch.randn(N, A, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.02
    U = K.wino_weights(w, x3=True)
    out = torch.empty(N, A, Cout, device="cuda")
    t = []
    for on in (1, 0):
        lib.erd_wino_x3_wide(on)
        t.append(timeit(lambda: K.wino_conv3x3(level_views(x, sizes), U, level_views(out, sizes), Cout)))
    lib.erd_wino_x3_wide(1)
    flop = 2.0 * N * A * Cin * Cout * 9
    print("%-32s %10.1f %10.1f   (%.0f / %.0f alg TF)" % (name, t[0], t[1], flop / t[0] / 1e6, flop / t[1] / 1e6))
