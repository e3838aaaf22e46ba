"""Order-dependence probe for the Winograd input-gradient launch of test_conv_dgrad_wgrad[1-256-80-13-21-3-1-1]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import golden_inputs as G
from erd_amd import kernels as K

def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().cuda()
def to_nchw(t): return t.permute(0, 3, 1, 2).cpu()
def relerr(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-12))

def case(N, Cin, Cout, H, W, k, s, p, tag):
    x = G.randn(11, N, Cin, H, W).requires_grad_(True)
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = G.randn(13, *y.shape)
    y.backward(dy)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wt1 = K.weight_transpose(wg, None)
    dx = torch.zeros((N, H, W, Cin), device="cuda")
    K.conv_dgrad([nhwc(dy)], wt1, [dx], k, s, p)
    torch.cuda.synchronize()
    print(tag, "plain", relerr(to_nchw(dx), x.grad), "sched", [t.tolist() for t in K._WINO_SCHED.values()])
    base = G.randn(15, N, Cin, H, W)
    dx2 = nhwc(base)
    K.conv_dgrad([nhwc(dy)], wt1, [dx2], k, s, p, accumulate=True)
    torch.cuda.synchronize()
    e = (to_nchw(dx2) - (x.grad + base)).abs()
    bad = (e > 1e-3).nonzero()
    print(tag, "accumulate", relerr(to_nchw(dx2), x.grad + base), "bad", len(bad), "first", bad[:3].tolist(), "last", bad[-3:].tolist(),
          "sched", [t.tolist() for t in K._WINO_SCHED.values()])

if "pre" in sys.argv:
    case(2, 128, 128, 26, 30, 3, 1, 1, "pre ")
case(1, 256, 80, 13, 21, 3, 1, 1, "A   ")
case(1, 256, 80, 13, 21, 3, 1, 1, "B   ")
