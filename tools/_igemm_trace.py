import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
N = 4
SH = [("L3.conv1", 1024, 256, 50, 84, 1, 1), ("L3.conv3", 256, 1024, 50, 84, 1, 1), ("L2.conv3", 128, 512, 100, 168, 1, 1),
      ("L1.conv3", 64, 256, 200, 336, 1, 1), ("L4.conv1", 2048, 512, 25, 42, 1, 1), ("L3.conv2s2", 256, 256, 100, 168, 3, 2)]
lib = _lib.load()
lib.erd_igemm_trace.argtypes = [C.c_void_p]
for name, Cin, Cout, H, W, k, s in SH:
    p = k // 2
    OH, OW = K.conv_out_size(H, k, s, p), K.conv_out_size(W, k, s, p)
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, OH, OW, Cout, device="cuda")
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    f = lambda: K.conv_forward([x], w, [y], k, s, p, scale=sc, shift=sh, relu=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); f(); e0.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8192)()
    lib.erd_igemm_trace(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    st, en = t[:, 0] - t0, t[:, 1] - t0
    tot = en - st
    print(f"{name}: {len(t)} wgs, event {s0.elapsed_time(e0)*1e3:.1f} us | start skew mean {st.mean():.0f} max {st.max():.0f} cyc | end min {en.min():.0f} mean {en.mean():.0f} max {en.max():.0f} | "
          f"per wg: total {tot.mean():.0f}, prologue {t[:,2].mean():.0f}, kloop {t[:,3].mean():.0f} ({t[:,6].mean():.1f} slices, {t[:,3].sum()/max(t[:,6].sum(),1):.0f} cyc/slice), fixup {t[:,4].mean():.0f}, epilogue {t[:,5].mean():.0f}")
