#!/usr/bin/env python3
"""How sensitive are the one-dispatch-round launches to CUs that are already taken (an RCCL kernel resident on
a few CUs pushes as many workgroups of a weight-gradient launch into a ragged second round)?  A dummy kernel occupies
`k` half-CUs on a side stream while the layer runs on the main stream; k = 0 is the undisturbed time.
usage: python tools/bench_wgrad_sensitivity.py   (needs hipcc on the box: compiles tools/occupy_cus.hip into /tmp)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from erd_amd import kernels as K

so = "/tmp/occupy_cus.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "occupy_cus.hip"),
                "-o", so], check=True)
lib = C.CDLL(so)
lib.occupy_cus.argtypes = [C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
side = torch.cuda.Stream()
sink = torch.zeros(1, dtype=torch.int64, device="cuda")
N = 4


def run(fn, k, iters=5):
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        if k:
            lib.occupy_cus(k, int(3e6), sink.data_ptr(), side.cuda_stream)      # ~1.4 ms of residency
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


def sizes_of(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out


cases = []
sizes = sizes_of(800, 1344)
A = sum(h * w for h, w in sizes)
x = torch.randn(N, A, 256, device="cuda"); dz = torch.randn(N, A, 256, device="cuda"); w3 = torch.randn(256, 3, 3, 256, device="cuda") * 0.05
y = torch.empty(N, A, 256, device="cuda")
xs, zs, ys = K.level_views(x, sizes), K.level_views(dz, sizes), K.level_views(y, sizes)
dW = torch.empty_like(w3)
def wg3():
    part, S = K.conv_wgrad_partials(xs, zs, 3, 1, 1); K.wgrad_reduce(part, S, w3, None, dW, False, None)
U = K.wino_weights(w3)
cases.append(("head-tower wgrad 3x3 (three-tap, one round)", wg3))
cases.append(("head-tower forward 3x3 (Winograd, claimed items)", lambda: K.wino_conv3x3(xs, U, ys, 256)))
x1 = torch.randn(N, 50, 84, 1024, device="cuda"); w1 = torch.randn(256, 1, 1, 1024, device="cuda") * 0.05; y1 = torch.empty(N, 50, 84, 256, device="cuda")
dz1 = torch.randn(N, 50, 84, 256, device="cuda"); dW1 = torch.empty_like(w1)
def wg1():
    part, S = K.conv_wgrad_partials([x1], [dz1], 1, 1, 0); K.wgrad_reduce(part, S, w1, None, dW1, False, None)
cases.append(("L3.conv1 wgrad 1x1 (generic, one round)", wg1))
cases.append(("L3.conv1 forward 1x1 (stream-K)", lambda: K.conv_forward([x1], w1, [y1], 1, 1, 0)))
cases.append(("L3.conv1 forward 1x1 (tile-parallel, no stream-K)", lambda: K.conv_forward([x1], w1, [y1], 1, 1, 0)))
print(f"{'launch':58s} " + " ".join(f"{'k=%d' % k:>9s}" for k in (0, 4, 16, 32)) + "   (us; k half-CUs occupied for the whole launch)")
for name, fn in cases:
    K.STREAMK = "no stream-K" not in name
    fn(); torch.cuda.synchronize()
    print(f"{name:58s} " + " ".join(f"{run(fn, k):9.1f}" for k in (0, 4, 16, 32)))
