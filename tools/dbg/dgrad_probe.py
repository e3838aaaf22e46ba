"""Scratch: 1x1 input-gradient launches (GEMM K = Cout of the forward conv), time + HBM rate; with a -DERD_IGEMM_TRACE library
(tools/build_probe.sh igtrace conv_mfma.hip -DERD_IGEMM_TRACE) also the per-workgroup phase cycles: DESIGN 7a."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from erd_amd import kernels as K, _lib
N = 4
# (name, Cin, Cout of the FORWARD conv, H, W): the input gradient is a GEMM with K = Cout, N = Cin
SH = [("L2.conv3 128->512 @100x168", 128, 512, 100, 168), ("L3.conv3 256->1024 @50x84", 256, 1024, 50, 84),
      ("L2.conv1 512->128 @100x168", 512, 128, 100, 168), ("L3.conv1 1024->256 @50x84", 1024, 256, 50, 84),
      ("L1.conv3 64->256 @200x336", 64, 256, 200, 336), ("L4.conv3 512->2048 @25x42", 512, 2048, 25, 42)]
lib = _lib.load()
trace = hasattr(lib, "erd_igemm_trace")
if trace: lib.erd_igemm_trace.argtypes = [C.c_void_p]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, Cin, Cout, H, W in SH:
    w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    wt = K.weight_transpose(w)
    dy = torch.randn(N, H, W, Cout, device="cuda"); dx = torch.zeros(N, H, W, Cin, device="cuda")
    f = lambda: K.conv_dgrad([dy], wt, [dx], 1, 1, 0)
    t = timeit(f)
    fl = 2.0 * N * H * W * Cin * Cout
    line = f"{name}: dgrad {t:6.1f} us {fl/t/1e6:5.0f} TF (HBM bytes {(dy.numel()+dx.numel())*4/1e6:.0f} MB -> {(dy.numel()+dx.numel())*4/t/1e6:.2f} TB/s)"
    if trace:
        f(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8192)(); lib.erd_igemm_trace(buf)
        tr = np.array(buf[:], dtype=np.float64).reshape(1024, 8); tr = tr[tr[:, 1] > 0]
        tot = tr[:, 1] - tr[:, 0]
        line += f" | {len(tr)} wgs: total {tot.mean():.0f} (max {tot.max():.0f}), prologue {tr[:,2].mean():.0f}, kloop {tr[:,3].mean():.0f} ({tr[:,6].mean():.1f} slices), fixup {tr[:,4].mean():.0f}, epilogue {tr[:,5].mean():.0f}; span {(tr[:,1].max()-tr[:,0].min()):.0f} cyc"
    print(line)
