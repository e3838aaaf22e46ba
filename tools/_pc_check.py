"""Scratch: the producer / consumer three-limb kernel (ERD_X3_PC=1) against the stream-K three-limb kernel on thin 1x1 layers:
results (forward with scale / shift / residual / ReLU; input gradient with mask / accumulate / column sums) and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erd_amd import kernels as K
N = 4
SH = [("L2.conv3", 128, 512, 100, 168), ("L3.conv3", 256, 1024, 50, 84), ("L1.conv3", 64, 256, 200, 336),
      ("L1.conv1b", 256, 64, 200, 336), ("L2.conv1b", 512, 128, 100, 168), ("ragged", 96, 200, 37, 53)]
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
def both(fn, get):
    os.environ["ERD_X3_PC"] = "0"; fn(); a = get(); ta = timeit(fn)
    os.environ["ERD_X3_PC"] = "1"; fn(); b = get(); tb = timeit(fn)
    os.environ["ERD_X3_PC"] = "0"
    return a, b, ta, tb
os.environ.setdefault("ERD_X3_PC_ROUNDS", "1")
torch.manual_seed(0)
for name, Cin, Cout, H, W in SH:
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5; sh = torch.rand(Cout, device="cuda"); r = torch.randn(N, H, W, Cout, device="cuda")
    y = torch.empty(N, H, W, Cout, device="cuda")
    def fwd():
        K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True, res=[r])
    def fwd_plain():
        K.conv_forward([x], w, [y], 1, 1, 0, scale=sc, shift=sh, relu=True)
    _, _, tpa, tpb = both(fwd_plain, lambda: None)
    a, b, ta, tb = both(fwd, lambda: y.clone())
    ref = torch.relu((x.double().reshape(-1, Cin) @ w.double().reshape(Cout, Cin).t()) * sc.double() + sh.double() + r.double().reshape(-1, Cout))
    ea = ((a.double().reshape(-1, Cout) - ref).abs().max() / ref.abs().max()).item(); eb = ((b.double().reshape(-1, Cout) - ref).abs().max() / ref.abs().max()).item()
    line = f"{name:10s} {Cin}->{Cout} @{H}x{W}: fwd plain {tpa:6.1f} -> {tpb:6.1f}, +res {ta:6.1f} -> {tb:6.1f} us, max|a-b| {(a-b).abs().max().item():.2e}, vs fp64 {ea:.1e} / {eb:.1e}"
    # input gradient of the same conv: K = Cout, N = Cin; mask, accumulate, column sums
    wt = K.weight_transpose(w)
    dy = torch.randn(N, H, W, Cout, device="cuda"); mk = torch.randn(N, H, W, Cin, device="cuda")
    dx = torch.empty(N, H, W, Cin, device="cuda"); cs = torch.zeros(Cin, device="cuda")
    def dgrad():
        cs.zero_()
        K.conv_dgrad([dy], wt, [dx], 1, 1, 0, res=[mk], relu_mask=[mk], colsum=cs)
    try:
        a, b, ta, tb = both(dgrad, lambda: torch.cat([dx.reshape(-1), cs]))
        refd = (dy.double().reshape(-1, Cout) @ w.double().reshape(Cout, Cin) + mk.double().reshape(-1, Cin)) * (mk.double().reshape(-1, Cin) > 0)
        refd = torch.cat([refd.reshape(-1), refd.sum(0)])
        ea = ((a.double() - refd).abs().max() / refd[:-Cin].abs().max()).item(); eb = ((b.double() - refd).abs().max() / refd[:-Cin].abs().max()).item()
        line += f" | dgrad {ta:6.1f} -> {tb:6.1f} us, max|a-b| {(a-b).abs()[:-Cin].max().item():.2e}, colsum rel {((a-b).abs()[-Cin:].max()/a.abs()[-Cin:].max()).item():.1e}, vs fp64 {ea:.1e} / {eb:.1e}"
    except Exception as e:
        line += f" | dgrad failed: {e}"
    print(line)
