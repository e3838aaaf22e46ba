#!/usr/bin/env python3
"""the bf16 mode's thin 1x1 launches: conv_thin_bf16_kernel against the stream-K kernel's bf16 instantiation (erd_conv_thin_enable 1 / 0), us per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K, _lib
K.set_compute("bf16")
lib = _lib.load()
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
N = 4
for name, Cin, Cout, H, W in [("L2.conv3 128->512 100x168", 128, 512, 100, 168), ("L1.conv3 64->256 200x336", 64, 256, 200, 336),
                               ("L1.conv1 64->64 200x336", 64, 64, 200, 336), ("L2.conv1 dgrad 128->512", 128, 512, 100, 168),
                               ("L3.conv3 256->1024 50x84", 256, 1024, 50, 84), ("L2.conv1_0 256->128 200x336", 256, 128, 200, 336),
                               ("L3.conv1 dgrad 256->1024", 256, 1024, 50, 84), ("fpn.lat2 256->256 100x168", 256, 256, 100, 168),
                               ("L2.conv1 512->128 100x168", 512, 128, 100, 168), ("L4.conv3 512->2048 25x42", 512, 2048, 25, 42),
                               ("fpn.lat3 512->256 100x168", 512, 256, 100, 168)]:
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.bfloat16); r = torch.randn_like(y)
    sc, sh = torch.rand(Cout, device="cuda"), torch.rand(Cout, device="cuda")
    row = []
    for on in (1, 0, 1, 0):
        lib.erd_conv_thin_enable(on)
        row.append(timeit(lambda: K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, res=[r], relu=True)))
    lib.erd_conv_thin_enable(1)
    mb = (x.numel() + 2 * y.numel()) * 2 / 1e6
    print("%-28s thin %6.1f / %6.1f us   stream-K %6.1f / %6.1f us   (%.0f MB: %.1f us at 8 TB/s)" % (name, row[0], row[2], row[1], row[3], mb, mb / 8.0))
