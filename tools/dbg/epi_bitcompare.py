#!/usr/bin/env python3
"""Do two builds of the library compute the same bits in every epilogue form of the three-limb conv kernels?
    python tools/dbg/epi_bitcompare.py OUT.pt            (ERD_HIP_LIB selects the build)
    python tools/dbg/epi_bitcompare.py OUT2.pt OUT.pt    compares while writing
Forms: forward (scale, shift, ReLU; + residual), input gradient (plain; mask; mask + accumulate; + column sums), on the implicit GEMM
(1x1 with K = 256 / 1024, Cout 80 / 128 / 512 / 1024; ragged pixel counts; two segments), the thin-K kernel (K = 64 / 128) and the
Winograd kernels (items of 64 and of 128 couts).  Column sums are compared with a tolerance (atomics), everything else bit for bit.
Timing of each form is printed too (microseconds, best of 3 x 10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from erd_amd import kernels as K

torch.manual_seed(0)
out, times = {}, {}


def timeit(fn, iters=10):
    best = 1e30
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best


def case(name, N, Cin, Cout, sizes, k, time_it=True):
    """sizes: list of (H, W): one segment each"""
    p = k // 2
    xs = [torch.randn(N, H, W, Cin, device="cuda") for H, W in sizes]
    w = torch.randn(Cout, k, k, Cin, device="cuda") * (2.0 / (k * k * Cin)) ** 0.5
    wt = K.weight_transpose(w)
    sc, sh = 0.5 + torch.rand(Cout, device="cuda"), 0.1 * torch.randn(Cout, device="cuda")
    ys = [torch.empty(N, H, W, Cout, device="cuda") for H, W in sizes]
    ident = [torch.randn_like(y) for y in ys]
    dys = [torch.randn_like(y) for y in ys]
    masks = [torch.randn_like(x) for x in xs]
    base = [torch.randn_like(x) for x in xs]
    res = {}

    def run(tag, fn, outs, tol=None):
        for o in outs: o.fill_(float("nan")) if tag.startswith("fwd") or tag == "dgrad" or tag == "dgrad_mask" else None
        fn(); torch.cuda.synchronize()
        res[tag] = [o.detach().cpu().clone() for o in outs]
        if time_it: times[name + "/" + tag] = timeit(fn)

    run("fwd", lambda: K.conv_forward(xs, w, ys, k, 1, p, scale=sc, shift=sh, relu=True), ys)
    run("fwd_res", lambda: K.conv_forward(xs, w, ys, k, 1, p, scale=sc, shift=sh, res=ident, relu=True), ys)
    dxs = [torch.empty_like(x) for x in xs]
    run("dgrad", lambda: K.conv_dgrad(dys, wt, dxs, k, 1, p), dxs)
    run("dgrad_mask", lambda: K.conv_dgrad(dys, wt, dxs, k, 1, p, relu_mask=masks), dxs)
    acc = [b.clone() for b in base]

    def acc_fn():
        for a, b in zip(acc, base): a.copy_(b)
        K.conv_dgrad(dys, wt, acc, k, 1, p, accumulate=True, relu_mask=masks)
    acc_fn(); torch.cuda.synchronize()
    res["dgrad_mask_acc"] = [a.cpu().clone() for a in acc]
    if time_it:
        t_copy = timeit(lambda: [a.copy_(b) for a, b in zip(acc, base)])
        times[name + "/dgrad_mask_acc"] = timeit(acc_fn) - t_copy
    cs = torch.zeros(8, Cin, device="cuda")
    K.conv_dgrad(dys, wt, dxs, k, 1, p, relu_mask=masks, colsum=cs); torch.cuda.synchronize()
    res["dgrad_mask_colsum"] = [d.cpu().clone() for d in dxs]
    res["colsum~"] = [cs.sum(0).cpu()]
    out[name] = res


def case_s2(name, N, Cin, Cout, H, W, k):
    """a stride-2 convolution through the implicit GEMM: forward (padding taps) and the merged-parity-class input gradient"""
    p = k // 2
    OH, OW = K.conv_out_size(H, k, 2, p), K.conv_out_size(W, k, 2, p)
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k, k, Cin, device="cuda") * (2.0 / (k * k * Cin)) ** 0.5
    wt = K.weight_transpose(w)
    sc, sh = 0.5 + torch.rand(Cout, device="cuda"), 0.1 * torch.randn(Cout, device="cuda")
    y = torch.full((N, OH, OW, Cout), float("nan"), device="cuda")
    K.conv_forward([x], w, [y], k, 2, p, scale=sc, shift=sh, relu=True); torch.cuda.synchronize()
    dy = torch.randn_like(y)
    dx = torch.full_like(x, float("nan"))
    K.conv_dgrad([dy], wt, [dx], k, 2, p); torch.cuda.synchronize()
    out[name] = {"fwd": [y.cpu()], "dgrad": [dx.cpu()]}
    times[name + "/fwd"] = timeit(lambda: K.conv_forward([x], w, [y], k, 2, p, scale=sc, shift=sh, relu=True))
    times[name + "/dgrad"] = timeit(lambda: K.conv_dgrad([dy], wt, [dx], k, 2, p))


case("igemm 256->128 37x53", 2, 256, 128, [(37, 53)], 1)
case("igemm 64->64 3x3 57x75", 2, 64, 64, [(57, 75)], 3)
case("igemm 68->256 3x3 levels", 2, 68, 256, [(25, 42), (13, 21)], 3, time_it=False)
case_s2("igemm 128->128 3x3 s2 100x168", 4, 128, 128, 100, 168, 3)
case_s2("igemm 256->256 3x3 s2 25x42", 4, 256, 256, 25, 42, 3)
case_s2("igemm 256->512 1x1 s2 100x168", 4, 256, 512, 100, 168, 1)
case("igemm 256->1024 50x84", 4, 256, 1024, [(50, 84)], 1)
case("igemm 1024->256 50x84", 4, 1024, 256, [(50, 84)], 1)
case("igemm 512->128 100x168", 4, 512, 128, [(100, 168)], 1)
case("igemm 256->80 two levels", 2, 256, 80, [(25, 42), (13, 21)], 1, time_it=False)
case("igemm 512->2048 25x42", 4, 512, 2048, [(25, 42)], 1)
case("thin 64->256 200x336", 4, 64, 256, [(200, 336)], 1)
case("thin 128->512 100x168", 4, 128, 512, [(100, 168)], 1)
case("thin 128->256 ragged", 1, 128, 256, [(19, 23)], 1, time_it=False)
os.environ["ERD_WINO_P"] = "2"
case("wino128 256->256 100x168", 4, 256, 256, [(100, 168)], 3)
case("wino128 128->128 100x168", 4, 128, 128, [(100, 168)], 3)
case("wino128 256->256 levels", 2, 256, 256, [(25, 42), (13, 21), (7, 11)], 3, time_it=False)
os.environ["ERD_WINO_P"] = "0"
case("wino64 256->256 50x84", 4, 256, 256, [(50, 84)], 3)
# the 7x7 / 2 stem (conv + folded BN + ReLU)
xs_ = torch.randn(4, 3, 800, 1344, device="cuda"); ws_ = torch.randn(64, 7, 7, 3, device="cuda") * 0.1
scs, shs = 0.5 + torch.rand(64, device="cuda"), 0.1 * torch.randn(64, device="cuda")
ys_ = K.stem(xs_, ws_, scs, shs); torch.cuda.synchronize()
out["stem 4x800x1344"] = {"fwd": [ys_.cpu()]}
times["stem 4x800x1344/fwd"] = timeit(lambda: K.stem(xs_, ws_, scs, shs))
torch.save(out, sys.argv[1])
for k_, v in times.items(): print(f"  {k_:46s} {v:8.1f} us")
if len(sys.argv) > 2:
    a = torch.load(sys.argv[2])
    bad = 0
    for n in a:
        for tag in a[n]:
            for u, v in zip(a[n][tag], out[n][tag]):
                if tag.endswith("~"):
                    ok = bool(((u - v).abs().max() / (u.abs().max() + 1e-30)) < 1e-5)
                else:
                    ok = bool(torch.equal(u, v)) and not bool(torch.isnan(v).any())
                bad += not ok
                if not ok: print("MISMATCH", n, tag, float((u - v).abs().max()), "nan" if bool(torch.isnan(v).any()) else "")
    print("all forms equal" if not bad else f"{bad} mismatches")
    sys.exit(1 if bad else 0)
