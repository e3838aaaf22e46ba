#!/usr/bin/env python3
"""Which torch-CPU operator of the bottleneck reference (tests/test_gpu_functions.py::test_fused_bottleneck_block[1024-512-2-True]) changes
its fp32 result with the thread count on this host?  Every stage's output and every gradient at nt threads against the same at 128."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.nn.functional as F
import golden_inputs as G

N, H, W, cin, planes, stride = 2, 12, 14, 1024, 512, 2
x0 = G.randn(90, N, cin, H, W)
P = {"w1": G.randn(1, planes, cin, 1, 1, scale=(2.0 / cin) ** 0.5), "w2": G.randn(2, planes, planes, 3, 3, scale=(2.0 / (9 * planes)) ** 0.5),
     "w3": G.randn(3, 4 * planes, planes, 1, 1, scale=(2.0 / planes) ** 0.5), "wd": G.randn(5, 4 * planes, cin, 1, 1, scale=(2.0 / cin) ** 0.5)}
for i, (n, c) in enumerate((("1", planes), ("2", planes), ("3", 4 * planes), ("d", 4 * planes))):
    P["g" + n], P["b" + n] = 0.5 + G.rand(10 + i, c), G.randn(20 + i, c, scale=0.2)
    P["m" + n], P["v" + n] = G.randn(30 + i, c, scale=0.2), 0.5 + G.rand(40 + i, c)
dy = G.randn(91, N, 4 * planes, 6, 7)


def run(nt, dtype=torch.float32):
    torch.set_num_threads(nt)
    p = {k: v.to(dtype).clone().requires_grad_(k[0] in "wgb") for k, v in P.items()}
    x = x0.to(dtype).clone().requires_grad_(True)
    bn = lambda t, n: F.batch_norm(t, p["m" + n], p["v" + n], p["g" + n], p["b" + n], False, 0.0, 1e-5)
    t = {}
    t["c1"] = F.conv2d(x, p["w1"]); t["r1"] = F.relu(bn(t["c1"], "1"))
    t["c2"] = F.conv2d(t["r1"], p["w2"], None, stride, 1); t["r2"] = F.relu(bn(t["c2"], "2"))
    t["c3"] = F.conv2d(t["r2"], p["w3"]); t["o"] = bn(t["c3"], "3")
    t["cd"] = F.conv2d(x, p["wd"], None, stride); t["idn"] = bn(t["cd"], "d")
    t["y"] = F.relu(t["o"] + t["idn"])
    for v in t.values(): v.retain_grad()
    t["y"].backward(dy.to(dtype))
    out = {"fwd " + k: v.detach().double() for k, v in t.items()}
    out.update({"grad " + k: v.grad.double() for k, v in t.items() if v.grad is not None})
    out["grad x"] = x.grad.double()
    out.update({"grad " + k: v.grad.double() for k, v in p.items() if v.grad is not None})
    return out


ref = run(128)
for nt in (16, 32, 64):
    g = run(nt)
    bad = {k: float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)) for k in ref}
    print("%d threads vs 128: " % nt + "  ".join("%s %.1e" % (k, v) for k, v in bad.items() if v > 1e-5) + " | everything else <= 1e-5", flush=True)
