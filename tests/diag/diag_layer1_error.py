#!/usr/bin/env python3
"""layer1's 3x3 convolutions (64->64 on 200x336) on the real activations of the procedural network: direct vs Winograd
against fp64"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from oracle import erd_oracle as O
from e2e_util import f7_state_dicts
from erd_amd import kernels as K

tsd, _ = f7_state_dicts()
imgs, _, _ = O.synthetic_batch(1, 800, 1333, 40, seed=7)
x, _ = O.preprocess(imgs)
sd = tsd
def bn(t, p):
    return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
with torch.no_grad():
    h = F.max_pool2d(F.relu(bn(F.conv2d(x, sd["backbone.conv1.weight"], None, 2, 3), "backbone.bn1")), 3, 2, 1)
    for b in range(3):
        p = f"backbone.layer1.{b}"
        o1 = F.relu(bn(F.conv2d(h, sd[p + ".conv1.weight"]), p + ".bn1"))
        w = sd[p + ".conv2.weight"]
        ref = F.conv2d(o1.double(), w.double(), None, 1, 1)
        xg = o1.permute(0, 2, 3, 1).contiguous().cuda(); wg = w.permute(0, 2, 3, 1).contiguous().cuda()
        out = torch.empty(1, o1.shape[2], o1.shape[3], 64, device="cuda")
        rel = lambda y: float((y.permute(0, 3, 1, 2).cpu().double() - ref).norm() / ref.norm())
        mx = lambda y: float((y.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
        K.WINOGRAD = False; K.conv_forward([xg], wg, [out], 3, 1, 1); ed, md = rel(out), mx(out)
        K.WINOGRAD = True; K.wino_conv3x3([xg], K.wino_weights(wg), [out], 64); ew, mw = rel(out), mx(out)
        ec = float((F.conv2d(o1, w, None, 1, 1).double() - ref).norm() / ref.norm())
        print("layer1.%d.conv2: input mean %.2f std %.2f max %.1f | rel L2: direct %.2e Winograd %.2e torch-CPU %.2e | max-norm: direct %.2e Winograd %.2e"
              % (b, float(o1.mean()), float(o1.std()), float(o1.max()), ed, ew, ec, md, mw))
        o2 = F.relu(bn(F.conv2d(o1, w, None, 1, 1), p + ".bn2"))
        o3 = bn(F.conv2d(o2, sd[p + ".conv3.weight"]), p + ".bn3")
        idn = bn(F.conv2d(h, sd[p + ".downsample.0.weight"]), p + ".downsample.1") if b == 0 else h
        h = F.relu(o3 + idn)
