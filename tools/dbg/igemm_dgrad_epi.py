#!/usr/bin/env python3
"""What the output stage of the 1x1 input-gradient launches costs (three-limb mode): plain | mask | column sums | residual + mask + sums,
next to the forward launch of the same GEMM shape."""
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


K.set_compute(os.environ.get("MODE", "f32x3"))
# (name, channels of dz = contraction, channels of dx, H, W)
for name, Cz, Cx, (H, W) in (("L4.conv3 dgrad", 2048, 512, (25, 42)), ("L3.conv1 dgrad", 256, 1024, (50, 84)), ("L3.conv3 dgrad", 1024, 256, (50, 84)),
                             ("L2.conv3 dgrad", 512, 128, (100, 168)), ("L4.conv1 dgrad", 512, 2048, (25, 42))):
    dz = torch.randn(4, H, W, Cz, device="cuda")
    w = torch.randn(Cz, 1, 1, Cx, device="cuda") * 0.05            # forward weight [Cout = Cz][1][1][Cin = Cx]
    wt = K.weight_transpose(w, None)
    dx = torch.empty(4, H, W, Cx, device="cuda")
    m = torch.randn(4, H, W, Cx, device="cuda")
    r = torch.randn(4, H, W, Cx, device="cuda")
    wf = torch.randn(Cx, 1, 1, Cz, device="cuda") * 0.05           # a forward conv of the same GEMM shape: Cz -> Cx
    cp = K.colsum_copies(Cx)
    row = [f"{name} {Cz}->{Cx}"]
    row.append("fwd %.1f" % timeit(lambda: K.conv_forward([dz], wf, [dx], 1, 1, 0)))
    for label, kw in (("plain", {}), ("mask", dict(relu_mask=[m])), (f"colsum{cp}", dict(colsum=torch.zeros(cp, Cx, device="cuda"))),
                      ("res", dict(res=[r])), ("mask+colsum", dict(relu_mask=[m], colsum=torch.zeros(cp, Cx, device="cuda"))),
                      ("res+mask+colsum", dict(res=[r], relu_mask=[m], colsum=torch.zeros(cp, Cx, device="cuda")))):
        row.append("%s %.1f" % (label, timeit(lambda: K.conv_dgrad([dz], wt, [dx], 1, 1, 0, **kw))))
    print(" | ".join(row), flush=True)
