"""Scratch: forward implicit GEMM of four thin 1x1 layers, a K = 1024 one and a 3x3 one in the current compute mode
(ERD_COMPUTE), one line; used with ERD_IG_LDS_PAD (one workgroup per CU) for the occupancy sensitivity of DESIGN 7a."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K
N = 4
SH = [("L2.conv3 128->512 @100x168", 128, 512, 100, 168, 1), ("L3.conv3 256->1024 @50x84", 256, 1024, 50, 84, 1),
      ("L1.conv3 64->256 @200x336", 64, 256, 200, 336, 1), ("L3.conv1 1024->256 @50x84", 1024, 256, 50, 84, 1),
      ("L2.conv1 512->128 @100x168", 512, 128, 100, 168, 1), ("fpn.out3 3x3 256->256 @100x168", 256, 256, 100, 168, 3)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out = []
for name, Cin, Cout, H, W, k in SH:
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    if K.COMPUTE == "bf16": x = x.bfloat16()
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=x.dtype)
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.rand(Cout, device="cuda")
    t = timeit(lambda: K.conv_forward([x], w, [y], k, 1, k // 2, scale=sc, shift=sh, relu=True))
    out.append(f"{name.split()[0]} {t:6.1f}")
print(K.COMPUTE, "stagger", os.environ.get("ERD_IG_STAGGER", "0"), " | ".join(out))
