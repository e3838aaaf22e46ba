#!/usr/bin/env python3
"""Phase trace of conv_thin_x3_kernel (build: tools/build_probe.sh thintrace conv_thin.hip -DERD_THIN_TRACE; run with
ERD_HIP_LIB=erd_amd/lib/abl/liberd_hip_thintrace.so): cycles of wave 0 of every workgroup, summed per phase over its blocks."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from erd_amd import kernels as K, _lib

N = 4
torch.manual_seed(0)
lib = _lib.load()
has_trace = hasattr(lib, "erd_thin_trace")


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def trace(name):
    if not has_trace:
        return
    buf = (C.c_ulonglong * 8192)()
    lib.erd_thin_trace.argtypes = [C.c_void_p]
    lib.erd_thin_trace(buf)
    t = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
    t = t[t[:, 7] > 0]
    tot, nb = t[:, 0].mean(), t[:, 7].mean()
    m = t.mean(0)
    print(f"   {name}: {len(t)} workgroups, {nb:.1f} blocks each, {tot:.0f} cycles = {tot/nb:.0f} per block | tile prologue {m[1]/tot:.1%} | ring store + barrier {m[2]/tot:.1%} "
          f"| colsum/pf/scale issue {m[3]/tot:.1%} | MFMA loop {m[4]/tot:.1%} | acc->LDS + wait {m[5]/tot:.1%} | rows out {m[6]/tot:.1%} | rest {(tot-m[1:7].sum())/tot:.1%}", flush=True)


for name, Cin, Cout, H, W, mode in [("L1.conv3 fwd 64->256 200x336 bn+relu", 64, 256, 200, 336, "fwd"), ("L2.conv3 fwd 128->512 100x168 bn+relu", 128, 512, 100, 168, "fwd"),
                                    ("L2.conv1 dgrad 128->512 100x168 plain", 512, 128, 100, 168, "dgrad"), ("L2.conv1 dgrad 128->512 masked + accumulated", 512, 128, 100, 168, "dgrad_m"),
                                    ("L1.conv1 dgrad 64->256 200x336 masked", 256, 64, 200, 336, "dgrad_m")]:
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    y = torch.empty(N, H, W, Cout, device="cuda"); dy = torch.randn_like(y); dx = torch.zeros_like(x)
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    wt = K.weight_transpose(w)
    if mode == "fwd":
        fn = lambda: K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True)
        byts = x.numel() * 4 + y.numel() * 4
    elif mode == "dgrad":
        fn = lambda: K.conv_dgrad([dy], wt, [dx], 1, 1, 0)
        byts = x.numel() * 4 + y.numel() * 4
    else:
        msk = torch.randn_like(x)
        fn = lambda: K.conv_dgrad([dy], wt, [dx], 1, 1, 0, relu_mask=[msk], accumulate=True)
        byts = x.numel() * 12 + y.numel() * 4
    t = min(timeit(fn) for _ in range(3))
    if mode == "dgrad_m":
        dx.zero_()
    fn(); torch.cuda.synchronize()
    out = y if mode == "fwd" else dx
    print(f"   bits {int(out.view(torch.int32).long().sum())}", end=" ")
    print(f"{name}: {t:.1f} us, {byts/1e6:.0f} MB algorithmic = {byts/t/1e6:.2f} TB/s", flush=True)
    trace(name)
