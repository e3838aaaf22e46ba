#!/usr/bin/env python3
"""Is torch-CPU's fp32 convolution backward on THIS host independent of the thread count?  (tests/conftest.py, round 5: a session-wide cap of
16 threads made tests/test_gpu_functions.py::test_fused_bottleneck_block[1024-512-2-True] read 1.3e-2 between the HIP block and its
torch-CPU reference on the GPU box's EPYC 9575F -- with BOTH libraries -- while the uncapped reference agrees to 1e-5.)  Prints the relative
L2 distance of the fp32 input gradient of each piece of that block to its fp64 evaluation, per thread count."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.nn.functional as F
import golden_inputs as G

N, H, W, cin, planes = 2, 12, 14, 1024, 512
x0 = G.randn(90, N, cin, H, W)
w1 = G.randn(1, planes, cin, 1, 1, scale=(2.0 / cin) ** 0.5)
w2 = G.randn(2, planes, planes, 3, 3, scale=(2.0 / (9 * planes)) ** 0.5)
w3 = G.randn(3, 4 * planes, planes, 1, 1, scale=(2.0 / planes) ** 0.5)
wd = G.randn(5, 4 * planes, cin, 1, 1, scale=(2.0 / cin) ** 0.5)
dy = G.randn(91, N, 4 * planes, 6, 7)


def grads(dtype, nt):
    torch.set_num_threads(nt)
    out = {}
    x = x0.to(dtype).clone().requires_grad_(True)
    F.conv2d(x, wd.to(dtype), None, 2).backward(dy.to(dtype)); out["1x1 stride 2 (downsample)"] = x.grad.double()
    x = x0.to(dtype).clone().requires_grad_(True)
    o = F.relu(F.conv2d(x, w1.to(dtype)))
    o = F.relu(F.conv2d(o, w2.to(dtype), None, 2, 1))
    F.conv2d(o, w3.to(dtype)).backward(dy.to(dtype)); out["1x1 -> 3x3 stride 2 -> 1x1 chain"] = x.grad.double()
    h = torch.randn(N, planes, H, W, generator=torch.Generator().manual_seed(3)).to(dtype).requires_grad_(True)
    F.conv2d(h, w2.to(dtype), None, 2, 1).backward(dy[:, :planes].to(dtype)); out["3x3 stride 2"] = h.grad.double()
    return out


print("torch", torch.__version__, "| default threads", torch.get_num_threads(), "| mkldnn", torch.backends.mkldnn.is_available())
ref = grads(torch.float64, min(torch.get_num_threads(), 32))
for nt in (1, 8, 16, 24, 32, 64, 128, 256):
    g = grads(torch.float32, nt)
    print("%3d threads: " % nt + "  ".join("%s %.1e" % (k, float((g[k] - ref[k]).norm() / ref[k].norm())) for k in ref), flush=True)
