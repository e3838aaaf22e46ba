import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
N = 4
SH = [("L3.conv1 1024->256 50x84", 1024, 256, 50, 84, 1, 1), ("L3.conv3 256->1024 50x84", 256, 1024, 50, 84, 1, 1), ("fpn.lat3 512->256 100x168", 512, 256, 100, 168, 1, 1),
      ("L2.conv1_0 256->128 200x336", 256, 128, 200, 336, 1, 1), ("L4.conv1 2048->512 25x42", 2048, 512, 25, 42, 1, 1), ("L4.conv3 512->2048 25x42", 512, 2048, 25, 42, 1, 1),
      ("L3.conv2 3x3 s2 256->256 100x168", 256, 256, 100, 168, 3, 2)]
lib = _lib.load()
lib.erd_igemm_trace.argtypes = [C.c_void_p]
def snapshot():
    buf = (C.c_ulonglong * 8192)()
    lib.erd_igemm_trace(buf)
    return np.array(buf[:], dtype=np.float64).reshape(1024, 8)


for name, Cin, Cout, H, W, k, s in SH:
    p = k // 2
    OH, OW = K.conv_out_size(H, k, s, p), K.conv_out_size(W, k, s, p)
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, OH, OW, Cout, device="cuda")
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    f = lambda: K.conv_forward([x], w, [y], k, s, p, scale=sc, shift=sh, relu=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    before = snapshot()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(); f(); e0.record(); torch.cuda.synchronize()
    after = snapshot()
    d = after                                  # (every workgroup resets its slots when it starts: conv_mfma.hip IG_SET)
    live = (after[:, 0] > before[:, 0].max()) & (after[:, 3] > 0)      # entries written by THIS launch (start stamps are monotonic)
    if not live.any():
        print(f"{name}: no implicit-GEMM workgroups traced (another kernel serves this shape)")
        continue
    t = d[live]
    tot = (after[live, 1] - after[live, 0])
    print(f"{name}: {int(live.sum())} wgs, event {s0.elapsed_time(e0)*1e3:.1f} us | per wg: total {tot.mean():.0f} ticks, row table + first loads {t[:,2].mean():.0f} ({t[:,2].sum()/tot.sum():.0%}), "
          f"K loop {t[:,3].mean():.0f} ({t[:,3].sum()/tot.sum():.0%}; {t[:,6].mean():.1f} slices, {t[:,3].sum()/max(t[:,6].sum(),1):.0f} per slice), stream-K fix-up {t[:,4].mean():.0f} ({t[:,4].sum()/tot.sum():.0%}), "
          f"epilogue {t[:,5].mean():.0f} ({t[:,5].sum()/tot.sum():.0%})", flush=True)
