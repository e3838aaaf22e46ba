"""Scratch: forward bf16 (bf16-stored maps) implicit GEMM on five shapes, one line; run once per probe library
(ERD_HIP_LIB=erd_amd/lib/abl/liberd_hip_NAME.so, built by tools/build_probe.sh NAME conv_mfma.hip -DERD_IG_NOMFMA ...): DESIGN 7a."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K
K.set_compute("bf16")
N = 4
SH = [("fpn.out3 3x3 256->256 @100x168", 256, 256, 100, 168, 3), ("gemm 2048^3 (1x1)", 2048, 2048, 16, 32, 1),
      ("L3.conv1 1024->256 @50x84", 1024, 256, 50, 84, 1), ("L2.conv3 128->512 @100x168", 128, 512, 100, 168, 1),
      ("L3.conv2 3x3 256->256 @50x84", 256, 256, 50, 84, 3)]
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
    n = N if "gemm" not in name else 4
    x = torch.randn(n, H, W, Cin, device="cuda").bfloat16(); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(n, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.rand(Cout, device="cuda")
    t = timeit(lambda: K.conv_forward([x], w, [y], k, 1, k // 2, scale=sc, shift=sh, relu=True))
    fl = 2.0 * n * H * W * Cout * Cin * k * k
    out.append(f"{name}: {t:7.1f} us {fl/t/1e6:6.0f} TF")
print(os.environ.get("ERD_HIP_LIB", "base").split("_")[-1], " | ".join(out))
