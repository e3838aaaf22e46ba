#!/usr/bin/env python3
"""where does the Winograd kernel's output error come from?  One 256->256 3x3 layer on a 100x168 map against an fp64 CPU
convolution: direct implicit GEMM, Winograd with the fp32 weight image, Winograd with a weight image computed in fp64
and rounded once."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from erd_amd import kernels as K

torch.manual_seed(0)
N, C, H, W = 1, 256, 100, 168
x = torch.randn(N, C, H, W).relu_()            # post-ReLU activations, as in the network
w = torch.randn(C, C, 3, 3) * (2.0 / (C * 9)) ** 0.5
ref = F.conv2d(x.double(), w.double(), None, 1, 1)
xg = x.permute(0, 2, 3, 1).contiguous().cuda()
wg = w.permute(0, 2, 3, 1).contiguous().cuda()
rel = lambda y: float((y.permute(0, 3, 1, 2).cpu().double() - ref).norm() / ref.norm())
out = torch.empty(N, H, W, C, device="cuda")
K.WINOGRAD = False
K.conv_forward([xg], wg, [out], 3, 1, 1)
print("direct implicit GEMM      rel L2 error %.3e" % rel(out))
K.WINOGRAD = True
U = K.wino_weights(wg)
K.wino_conv3x3([xg], U, [out], C)
print("Winograd, fp32 weight image %.3e" % rel(out))
# the same weight image computed in fp64 and rounded once: U = G g G^T per (co, ci), laid out as the kernel wants it
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
U64 = torch.einsum("ia,ocab,jb->ijoc", G, w.double(), G)            # [4,4,Cout,Cin]
# kernel layout: [16][Cout/32][Cin/4][32][4]
U64 = U64.reshape(16, C // 32, 32, C // 4, 4).permute(0, 1, 3, 2, 4).contiguous().float().cuda()
assert U64.numel() == U.numel()
print("max |U32 - U64| / max|U|: %.2e" % float((U.reshape(-1) - U64.reshape(-1)).abs().max() / U64.abs().max()))
K.wino_conv3x3([xg], U64.reshape(U.shape), [out], C)
print("Winograd, fp64-computed weight image %.3e" % rel(out))

# ---- the same on REAL activations: the teacher's first cls-tower layer on its own P3 features
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts
tsd, _ = f7_state_dicts()
imgs, _, _ = O.synthetic_batch(1, 800, 1333, 40, seed=7)
xin, _ = O.preprocess(imgs)
with torch.no_grad():
    feats = O.fpn_forward(tsd, O.resnet_forward(tsd, xin))
    p3 = feats[0]
    for layer in range(2):
        wt = tsd[f"bbox_head.cls_convs.{layer}.conv.weight"]
        ref = F.conv2d(p3.double(), wt.double(), None, 1, 1)
        xg = p3.permute(0, 2, 3, 1).contiguous().cuda()
        wg = wt.permute(0, 2, 3, 1).contiguous().cuda()
        out = torch.empty(1, p3.shape[2], p3.shape[3], 256, device="cuda")
        rel = lambda y: float((y.permute(0, 3, 1, 2).cpu().double() - ref).norm() / ref.norm())
        K.WINOGRAD = False
        K.conv_forward([xg], wg, [out], 3, 1, 1)
        e_d = rel(out)
        K.WINOGRAD = True
        K.wino_conv3x3([xg], K.wino_weights(wg), [out], 256)
        e_w = rel(out)
        e_c = float((F.conv2d(p3, wt, None, 1, 1).double() - ref).norm() / ref.norm())
        print("cls tower layer %d on real P3 features (mean/std of input %.3f/%.3f): direct %.3e  Winograd %.3e  torch-CPU fp32 %.3e"
              % (layer, float(p3.mean()), float(p3.std()), e_d, e_w, e_c))
        gnw, gnb = tsd[f"bbox_head.cls_convs.{layer}.gn.weight"], tsd[f"bbox_head.cls_convs.{layer}.gn.bias"]
        p3 = F.relu(F.group_norm(F.conv2d(p3, wt, None, 1, 1), 32, gnw, gnb))
