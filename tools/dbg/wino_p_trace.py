#!/usr/bin/env python3
"""Phase trace of wino_x3p_kernel (build: tools/build_abl.sh NAME; run with ERD_HIP_LIB=erd_amd/lib/abl/liberd_hip_NAME.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
os.environ["ERD_WINO_P"] = "2"
N = 4
def sizes_of(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out
for name, Cin, Cout, sizes in [("head tower", 256, 256, sizes_of(800, 1344)), ("L2.conv2", 128, 128, [(100, 168)])]:
    A = sum(h * w for h, w in sizes)
    x = torch.randn(N, A, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
    y = torch.empty(N, A, Cout, device="cuda")
    xs, ys = K.level_views(x, sizes), K.level_views(y, sizes)
    U = K.wino_weights(w, x3=True)
    for _ in range(3): K.wino_conv3x3(xs, U, ys, Cout)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 4096)()
    lib = _lib.load()
    lib.erd_wino_trace_p.argtypes = [C.c_void_p]
    lib.erd_wino_trace_p(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(256, 16)
    m = t.mean(0)
    nsl = (Cin // 16)
    for k, nm in ((0, "wave 0 (M then D)"), (8, "wave 4 (D then M)")):
        tot = m[k]
        print(f"{name} {nm}: total {tot:.0f} cyc | barrier {m[k+1]/tot:.1%} | M {m[k+2]/tot:.1%} | D {m[k+3]/tot:.1%} (reads+raw store {m[k+4]/tot:.1%}, "
              f"transform+V stores {m[k+5]/tot:.1%}, switch+ring tail {m[k+6]/tot:.1%}) | output {m[k+7]/tot:.1%} | rest {(tot-m[k+1]-m[k+2]-m[k+3]-m[k+7])/tot:.1%}")
