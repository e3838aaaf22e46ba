import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
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
    U = K.wino_weights(w)
    for _ in range(3): K.wino_conv3x3(xs, U, ys, Cout)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 2048)()
    lib = _lib.load()
    lib.erd_wino_trace.argtypes = [C.c_void_p]
    rc = lib.erd_wino_trace(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(256, 8)
    m = t.mean(0)
    x3 = U.dtype == torch.bfloat16      # (three-limb kernel: slot 3 holds the cycles spent polling the partner wave's exchange flag)
    print(f"{name} ({'three-limb' if x3 else 'fp32'}): MMA wave: total {m[0]:.0f} cyc, barrier wait {m[1]:.0f} ({m[1]/m[0]:.1%}), output stage {m[2]:.0f} ({m[2]/m[0]:.1%}), "
          f"{'exchange polls' if x3 else 'items'} {m[3]:.1f} | "
          f"data wave: total {m[4]:.0f}, barrier wait {m[5]:.0f} ({m[5]/m[4]:.1%}), store+issue {m[6]:.0f} ({m[6]/m[4]:.1%}), transform {m[7]:.0f} ({m[7]/m[4]:.1%})")
