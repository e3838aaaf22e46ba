"""Scratch: phase cycles of the producer / consumer three-limb kernel (library built with -DERD_PC_TRACE)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
os.environ["ERD_X3_PC"] = "1"; os.environ.setdefault("ERD_X3_PC_ROUNDS", "1")
lib = _lib.load(); lib.erd_pc_trace.argtypes = [C.c_void_p]
N = 4
for name, Cin, Cout, H, W, withres in [("L2.conv3", 128, 512, 100, 168, True), ("L2.conv3 plain", 128, 512, 100, 168, False), ("L3.conv3", 256, 1024, 50, 84, True), ("L1.conv3", 64, 256, 200, 336, True)]:
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.rand(Cout, device="cuda"); r = torch.randn(N, H, W, Cout, device="cuda")
    y = torch.empty(N, H, W, Cout, device="cuda")
    f = lambda: K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True, res=[r] if withres else None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); f(); e0.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 2048)(); lib.erd_pc_trace(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(256, 8)
    st = t[:, 3].mean()
    print(f"{name}: {s0.elapsed_time(e0)*1e3:.1f} us | matrix waves: total {t[:,0].mean():.0f} cyc, {st:.1f} steps, per step: multiply {t[:,2].mean()/st:.0f}, barrier {t[:,1].mean()/st:.0f} | data waves per step: store+issue {t[:,4].mean()/st:.0f}, epilogue {t[:,5].mean()/st:.0f}, barrier {t[:,6].mean()/st:.0f}")
