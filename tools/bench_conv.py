#!/usr/bin/env python3
"""per-shape TFLOP/s of the conv kernels (fwd / dgrad / wgrad) on the GFL-R50 layer shapes at bs=4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erd_amd import kernels as K

N = int(os.environ.get("BS", "4"))
SHAPES = [  # name, Cin, Cout, H, W, k, s
    ("L1.conv1_64", 64, 64, 200, 336, 1, 1), ("L1.conv1_256", 256, 64, 200, 336, 1, 1),
    ("L1.conv2", 64, 64, 200, 336, 3, 1), ("L1.conv3", 64, 256, 200, 336, 1, 1),
    ("L2.conv1_0", 256, 128, 200, 336, 1, 1), ("L2.conv2_s2", 128, 128, 200, 336, 3, 2),
    ("L2.conv1", 512, 128, 100, 168, 1, 1), ("L2.conv2", 128, 128, 100, 168, 3, 1),
    ("L2.conv3", 128, 512, 100, 168, 1, 1), ("L2.down", 256, 512, 200, 336, 1, 2),
    ("L3.conv1", 1024, 256, 50, 84, 1, 1), ("L3.conv2", 256, 256, 50, 84, 3, 1), ("L3.conv3", 256, 1024, 50, 84, 1, 1),
    ("L4.conv1", 2048, 512, 25, 42, 1, 1), ("L4.conv2", 512, 512, 25, 42, 3, 1), ("L4.conv3", 512, 2048, 25, 42, 1, 1),
    ("fpn.lat3", 512, 256, 100, 168, 1, 1), ("fpn.out3", 256, 256, 100, 168, 3, 1),
    ("head.P4", 256, 256, 50, 84, 3, 1), ("big.gemm", 2048, 2048, 128, 128, 1, 1), ("big.3x3", 256, 256, 200, 336, 3, 1), ("head.cls80", 256, 80, 100, 168, 3, 1), ("head.reg68", 256, 68, 100, 168, 3, 1),
]
only = sys.argv[1:]
for spec in os.environ.get("SHAPES", "").split(";"):      # extra shapes: name,Cin,Cout,H,W,k,s
    if spec:
        f = spec.split(",")
        SHAPES.append((f[0],) + tuple(int(v) for v in f[1:]))

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

print(f"{'layer':14s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
tot = [0, 0, 0, 0]
for name, Cin, Cout, H, W, k, s in SHAPES:
    if only and not any(o in name for o in only): continue
    p = k // 2
    OH, OW = K.conv_out_size(H, k, s, p), K.conv_out_size(W, k, s, p)
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, OH, OW, Cout, device="cuda"); dy = torch.randn_like(y); dx = torch.zeros_like(x)
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    wt = K.weight_transpose(w)
    fl = 2.0 * N * OH * OW * Cout * Cin * k * k
    t_f = timeit(lambda: K.conv_forward([x], w, [y], k, s, p, scale=sc, shift=sh, relu=True))
    t_d = timeit(lambda: K.conv_dgrad([dy], wt, [dx], k, s, p))
    dW = torch.empty_like(w)
    def wg():
        part, S = K.conv_wgrad_partials([x], [dy], k, s, p); K.wgrad_reduce(part, S, w, None, dW, False, None)
    t_w = timeit(wg)
    tot[0] += fl; tot[1] += t_f; tot[2] += t_d; tot[3] += t_w
    print(f"{name:14s} {fl/1e9:7.1f} | {t_f*1e3:8.1f} {fl/t_f/1e9:6.1f} | {t_d*1e3:8.1f} {fl/t_d/1e9:6.1f} | {t_w*1e3:8.1f} {fl/t_w/1e9:6.1f}")
print(f"{'sum':14s} {tot[0]/1e9:7.1f} | {tot[1]*1e3:8.1f} {tot[0]/tot[1]/1e9:6.1f} | {tot[2]*1e3:8.1f} {tot[0]/tot[2]/1e9:6.1f} | {tot[3]*1e3:8.1f} {tot[0]/tot[3]/1e9:6.1f}")
