#!/usr/bin/env python3
"""why does tools/bench_conv.py read 4.4 ms for the forward 128 -> 512 1x1 launch (r4c)?  time the pieces"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K, _lib

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

N, H, W = 4, 100, 168
for Cin, Cout in ((128, 512), (128, 256), (64, 512), (256, 512), (128, 1024)):
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    y = torch.empty(N, H, W, Cout, device="cuda")
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    for thin in (1, 0):
        _lib.load().erd_conv_thin_enable(thin)
        t_plain = timeit(lambda: K.conv_forward([x], w, [y], 1, 1, 0))
        t_epi = timeit(lambda: K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True))
        t_split = timeit(lambda: K.split3(w))
        print(f"{Cin}->{Cout} thin={thin}: plain {t_plain:8.1f} us   scale/shift/relu {t_epi:8.1f} us   split3(w) alone {t_split:6.1f} us", flush=True)
    K.set_compute("f32")
    t = timeit(lambda: K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True))
    print(f"{Cin}->{Cout} native f32: {t:8.1f} us")
    K.set_compute(K.DEFAULT_COMPUTE)
