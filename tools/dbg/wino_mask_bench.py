"""Winograd three-limb launches: what the epilogue variants cost (plain / ReLU mask / column sums / residual), us per launch"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from erd_amd import kernels as K
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, C, H, W in [("L2.conv2 128", 128, 100, 168), ("L3.conv2 256", 256, 50, 84), ("L4.conv2 512", 512, 25, 42), ("fpn P3 256", 256, 100, 168)]:
    x = torch.randn(4, H, W, C, device="cuda"); w = torch.randn(C, 3, 3, C, device="cuda") * 0.02
    U = K.wino_weights(w, x3=True)
    out = torch.empty_like(x); mask = torch.randn_like(x); res = torch.randn_like(x)
    cs = torch.zeros(int(os.environ.get("COPIES", "8")), C, device="cuda")
    t0 = timeit(lambda: K.wino_conv3x3([x], U, [out], C))
    t1 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, mask=[mask], colsum=cs))
    t2 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, res=[res], mask=[mask], colsum=cs))
    t3 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, res=[out]))
    t4 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, mask=[mask]))
    t5 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, colsum=cs))
    cs1 = torch.zeros(C, device="cuda")
    t6 = timeit(lambda: K.wino_conv3x3([x], U, [out], C, colsum=cs1))
    print("%-14s plain %6.1f | mask+colsum %6.1f | res+mask+colsum %6.1f | accumulate %6.1f | mask only %6.1f | colsum only (COPIES, default 8) %6.1f (1 copy) %6.1f us" % (name, t0, t1, t2, t3, t4, t5, t6))
