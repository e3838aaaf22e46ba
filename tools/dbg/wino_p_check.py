#!/usr/bin/env python3
"""wino_x3p_kernel (128 couts per item, phased waves) against wino_x3_kernel (64 couts per item, role-split waves): bit-equality
of every epilogue form on a set of shapes, then microseconds per launch of both on the 3x3 stride-1 shapes of the step.
ERD_WINO_P is read per launch (winograd.hip wino_launch): 0 = role-split kernel, 2 = phased kernel wherever Cout % 128 == 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K

torch.manual_seed(0)
NEW = int(os.environ.get("P_NEW", "2"))        # 2: wino_x3p_kernel, 3: wino_x3s_kernel


def run(mode, fn):
    os.environ["ERD_WINO_P"] = str(mode)
    fn()
    torch.cuda.synchronize()


def sizes_of(H, W):
    out, h, w = [], H // 8, W // 8
    for _ in range(5):
        out.append((h, w)); h, w = (h + 1) // 2, (w + 1) // 2
    return out


bad = 0
for (N, Cin, Cout, sizes) in [(2, 256, 256, [(26, 30)]), (1, 64, 128, [(20, 28)]), (2, 128, 128, [(25, 42)]), (1, 512, 512, [(7, 11)]),
                              (1, 64, 256, [(34, 66)]), (2, 256, 256, [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]),
                              (1, 256, 256, [(100, 168)]), (4, 256, 256, sizes_of(800, 1344))]:
    A = sum(h * w for h, w in sizes)
    x = torch.randn(N, A, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
    U = K.wino_weights(w, x3=True)
    scale, shift = 0.5 + torch.rand(Cout, device="cuda"), 0.1 * torch.randn(Cout, device="cuda")
    base, mask = torch.randn(N, A, Cout, device="cuda"), torch.randn(N, A, Cout, device="cuda")
    xs = K.level_views(x, sizes)
    outs = {}
    for mode in (0, NEW):
        y1 = torch.full((N, A, Cout), float("nan"), device="cuda")
        run(mode, lambda: K.wino_conv3x3(xs, U, K.level_views(y1, sizes), Cout))
        y2 = torch.full((N, A, Cout), float("nan"), device="cuda")
        run(mode, lambda: K.wino_conv3x3(xs, U, K.level_views(y2, sizes), Cout, scale=scale, shift=shift, relu=True))
        y3 = base.clone()
        cs = torch.zeros(8, Cout, device="cuda")
        v3 = K.level_views(y3, sizes)
        run(mode, lambda: K.wino_conv3x3(xs, U, v3, Cout, res=v3, mask=K.level_views(mask, sizes), colsum=cs))
        outs[mode] = (y1, y2, y3, cs.sum(0))
    eq = [bool(torch.equal(a, b)) for a, b in zip(outs[0][:3], outs[NEW][:3])]
    cs_rel = float((outs[0][3] - outs[NEW][3]).abs().max() / outs[0][3].abs().max())
    nan = [bool(torch.isnan(t).any()) for t in outs[NEW][:3]]
    ok = all(eq) and not any(nan) and cs_rel < 1e-5
    bad += not ok
    print(f"N{N} {Cin}->{Cout} {sizes[0]}x{len(sizes)}: plain/bn-relu/res-mask bit-equal {eq}, nan {nan}, colsum rel {cs_rel:.1e}  {'OK' if ok else 'MISMATCH'}",
          flush=True)

if os.environ.get("BENCH", "1") != "0":
    def timeit(fn, iters=10):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3

    N = 4
    CASES = [("head tower 5 levels", 256, 256, sizes_of(800, 1344)), ("fpn.out P3", 256, 256, [(100, 168)]),
             ("fpn.out P4", 256, 256, [(50, 84)]), ("L2.conv2", 128, 128, [(100, 168)]), ("L3.conv2", 256, 256, [(50, 84)]),
             ("L4.conv2", 512, 512, [(25, 42)])]
    print(f"{'launch':22s} {'GFLOP':>7s} | {'x3 us':>8s} {'alg TF':>7s} | {'x3p us':>8s} {'alg TF':>7s} | x3/x3p")
    for name, Cin, Cout, sizes in CASES:
        A = sum(h * w for h, w in sizes)
        x = torch.randn(N, A, Cin, device="cuda")
        w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
        y = torch.empty(N, A, Cout, device="cuda")
        xs, ys = K.level_views(x, sizes), K.level_views(y, sizes)
        U = K.wino_weights(w, x3=True)
        fl = 2.0 * N * A * Cout * Cin * 9
        t = {}
        for mode in (0, NEW, 0, NEW):
            os.environ["ERD_WINO_P"] = str(mode)
            t[mode] = min(t.get(mode, 1e30), timeit(lambda: K.wino_conv3x3(xs, U, ys, Cout)))
        print(f"{name:22s} {fl/1e9:7.1f} | {t[0]:8.1f} {fl/t[0]/1e6:7.1f} | {t[NEW]:8.1f} {fl/t[NEW]/1e6:7.1f} | {t[0]/t[NEW]:5.2f}x", flush=True)
sys.exit(1 if bad else 0)
