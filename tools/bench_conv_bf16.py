#!/usr/bin/env python3
"""bf16 matrix-core convolutions: fp32-stored maps against bf16-stored maps (erd_conv_desc::in_bf16 / out_bf16), per
GFL-R50 layer shape at bs=4, with the result checked against conv(bf16(x), bf16(w)) accumulated in fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from erd_amd import kernels as K

K.set_compute("bf16")
N = 4
SHAPES = [("L1.conv3", 64, 256, 200, 336, 1, 1), ("L2.conv1", 512, 128, 100, 168, 1, 1), ("L2.conv2", 128, 128, 100, 168, 3, 1),
          ("L2.conv3", 128, 512, 100, 168, 1, 1), ("L3.conv1", 1024, 256, 50, 84, 1, 1), ("L3.conv2", 256, 256, 50, 84, 3, 1),
          ("L3.conv3", 256, 1024, 50, 84, 1, 1), ("L4.conv2", 512, 512, 25, 42, 3, 1), ("fpn.out3", 256, 256, 100, 168, 3, 1),
          ("L2.conv2_s2", 128, 128, 200, 336, 3, 2), ("head.cls80", 256, 80, 100, 168, 3, 1)]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f"{'layer':12s} {'GFLOP':>6s} | fwd us: f32-stored  bf16-stored | dgrad us: f32-stored  bf16-stored | wgrad us: f32-stored bf16-stored | max err fwd (bf16 ulp) dgrad")
for name, Cin, Cout, H, W, k, s in SHAPES:
    p = k // 2
    OH, OW = K.conv_out_size(H, k, s, p), K.conv_out_size(W, k, s, p)
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, OH, OW, Cout, device="cuda"); dy = torch.randn_like(y); dx = torch.zeros_like(x)
    xb, yb, dyb, dxb = x.bfloat16(), y.bfloat16(), dy.bfloat16(), dx.bfloat16()
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.rand(Cout, device="cuda")
    wt = K.weight_transpose(w)
    fl = 2.0 * N * OH * OW * Cout * Cin * k * k
    t0 = timeit(lambda: K.conv_forward([x], w, [y], k, s, p, scale=sc, shift=sh, relu=True))
    t1 = timeit(lambda: K.conv_forward([xb], w, [yb], k, s, p, scale=sc, shift=sh, relu=True))
    t2 = timeit(lambda: K.conv_dgrad([dy], wt, [dx], k, s, p))
    t3 = timeit(lambda: K.conv_dgrad([dyb], wt, [dxb], k, s, p))
    t4 = timeit(lambda: K.conv_wgrad_partials([x], [dy], k, s, p))
    t5 = timeit(lambda: K.conv_wgrad_partials([xb], [dyb], k, s, p))
    # reference: products of bf16-rounded operands, fp32 accumulation, epilogue in fp32, one rounding on store
    ref = F.conv2d(xb.float().permute(0, 3, 1, 2), w.bfloat16().float().permute(0, 3, 1, 2), stride=s, padding=p)
    ref = torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    err_f = ((yb.float() - ref).abs() / ref.abs().clamp_min(1.0)).max().item() * 256
    refd = F.conv_transpose2d(dyb.float().permute(0, 3, 1, 2), w.bfloat16().float().permute(0, 3, 1, 2), stride=s, padding=p,
                              output_padding=(H + 2 * p - k) % s if s > 1 else 0)
    if refd.shape[2] != H or refd.shape[3] != W:
        refd = F.pad(refd, (0, W - refd.shape[3], 0, H - refd.shape[2]))
    refd = refd.permute(0, 2, 3, 1)
    err_d = ((dxb.float() - refd).abs() / refd.abs().clamp_min(1.0)).max().item() * 256
    print(f"{name:12s} {fl/1e9:6.1f} | {t0:10.1f} {t1:12.1f} ({fl/t1/1e6:5.0f} TF) | {t2:10.1f} {t3:12.1f} ({fl/t3/1e6:5.0f} TF) | {t4:8.1f} {t5:8.1f} ({fl/t5/1e6:5.0f} TF) | {err_f:8.2f} {err_d:8.2f}")
