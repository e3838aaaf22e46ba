#!/usr/bin/env python3
"""What the masked / column-summing output stage of the three-limb Winograd kernels costs per launch (the backbone's conv2 input gradients):
plain | mask | column sums (8 / 64 copies) | mask + column sums, on the three shapes of the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


K.set_compute("f32x3")
for name, C, (H, W) in (("L2.conv2", 128, (100, 168)), ("L3.conv2", 256, (50, 84)), ("L4.conv2", 512, (25, 42))):
    x = torch.randn(4, H, W, C, device="cuda")
    w = torch.randn(C, 3, 3, C, device="cuda") * 0.05
    y = torch.empty(4, H, W, C, device="cuda")
    m = torch.randn(4, H, W, C, device="cuda")
    U = K.wino_weights(w, x3=True)
    row = [name]
    for label, kw in (("plain", {}), ("mask", dict(mask=[m])), ("colsum8", dict(colsum=torch.zeros(8, C, device="cuda"))),
                      ("colsum64", dict(colsum=torch.zeros(64, C, device="cuda"))),
                      ("mask+colsum8", dict(mask=[m], colsum=torch.zeros(8, C, device="cuda"))),
                      ("mask+colsum64", dict(mask=[m], colsum=torch.zeros(64, C, device="cuda")))):
        t = timeit(lambda: K.wino_conv3x3([x], U, [y], C, **kw))
        row.append(f"{label} {t:.1f}")
    print(" | ".join(row), flush=True)
