"""outputs of a set of three-limb launches as a file (run once per library, then compare): do two builds compute the same bits?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ctypes as C
import erd_amd._lib as L
if os.environ.get("ERD_HIP_LIB"):          # (an older library -- another ABI version -- loaded by hand: only the launches below are used)
    lib = C.CDLL(os.environ["ERD_HIP_LIB"])
    lib.erd_last_error.restype = C.c_char_p
    for name, args in L._SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.argtypes = args
            fn.restype = C.c_size_t if name.endswith(("_ws_bytes", "_elems")) else C.c_int
    L._lib = lib
from erd_amd import kernels as K
torch.manual_seed(0)
out = {}
N = 2
for name, Cin, Cout, H, W, k, s in [("c1", 256, 128, 40, 56, 1, 1), ("c3", 128, 512, 40, 56, 1, 1), ("k3", 256, 256, 24, 40, 3, 1), ("s2", 128, 128, 40, 56, 3, 2), ("thin", 64, 256, 40, 56, 1, 1)]:
    p = k // 2
    OH, OW = K.conv_out_size(H, k, s, p), K.conv_out_size(W, k, s, p)
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    y = torch.empty(N, OH, OW, Cout, device="cuda"); dy = torch.randn_like(y); dx = torch.zeros_like(x)
    sc = torch.rand(Cout, device="cuda"); sh = torch.rand(Cout, device="cuda")
    wt = K.weight_transpose(w)
    K.conv_forward([x], w, [y], k, s, p, scale=sc, shift=sh, relu=True)
    K.conv_dgrad([dy], wt, [dx], k, s, p)
    dW = torch.empty_like(w)
    part, S = K.conv_wgrad_partials([x], [dy], k, s, p); K.wgrad_reduce(part, S, w, None, dW, False, None)
    torch.cuda.synchronize()
    out[name] = (y.cpu(), dx.cpu(), dW.cpu())
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    a, b = torch.load(sys.argv[2]), out
    for n in a:
        print(n, [bool(torch.equal(u, v)) for u, v in zip(a[n], b[n])], ["%.1e" % float((u - v).abs().max() / v.abs().max()) for u, v in zip(a[n], b[n])])
