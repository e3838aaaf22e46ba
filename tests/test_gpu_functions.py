"""GPU: each autograd.Function of erd_amd.functional (a fused group of HIP launches with a hand-written
backward) against torch-CPU autograd of the same composite op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def cl_weight(w):     # OIHW cpu -> channels_last parameter on the GPU
    return w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)


CBA = [
    # N,H,W,Cin,Cout,k,s,res,relu
    (2, 8, 10, 1024, 256, 1, 1, False, True),
    (2, 8, 10, 256, 256, 3, 1, False, True),
    (2, 8, 10, 256, 1024, 1, 1, True, True),
    (2, 8, 10, 1024, 2048, 1, 2, False, False),
    (2, 16, 20, 128, 128, 3, 2, False, True),
    (2, 7, 9, 512, 512, 3, 2, False, True),
    (2, 16, 20, 512, 128, 1, 1, False, True),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,res,relu", CBA)
def test_conv_bn_act(N, H, W, Cin, Cout, k, s, res, relu):
    from erd_amd import functional as Fn
    p = k // 2
    x = G.randn(1, N, Cin, H, W).requires_grad_(True)
    w = G.randn(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    gamma = (0.5 + G.rand(3, Cout)).requires_grad_(True)
    beta = G.randn(4, Cout, scale=0.2).requires_grad_(True)
    mean, var = G.randn(5, Cout, scale=0.2), 0.5 + G.rand(6, Cout)
    y = F.batch_norm(F.conv2d(x, w, None, s, p), mean, var, gamma, beta, False, 0.0, 1e-5)
    r = None
    if res:
        r = G.randn(7, *y.shape).requires_grad_(True)
        y = y + r
    if relu:
        y = F.relu(y)
    dy = G.randn(8, *y.shape)
    y.backward(dy)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wg = cl_weight(w.detach())
    gg, bg = gamma.detach().cuda().requires_grad_(True), beta.detach().cuda().requires_grad_(True)
    rg = r.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True) if res else None
    out = Fn.ConvBNAct.apply(xg, wg, gg, bg, mean.cuda(), var.cuda(), rg, k, s, p, relu, 1e-5)
    assert relerr(out.detach().permute(0, 3, 1, 2).cpu(), y.detach()) < 2e-5
    out.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    assert relerr(xg.grad.permute(0, 3, 1, 2).cpu(), x.grad) < 5e-5
    assert relerr(wg.grad.cpu(), w.grad) < 5e-5
    assert relerr(gg.grad.cpu(), gamma.grad) < 2e-4
    assert relerr(bg.grad.cpu(), beta.grad) < 5e-5
    if res:
        assert relerr(rg.grad.permute(0, 3, 1, 2).cpu(), r.grad) < 1e-6


def test_fpn_functions():
    from erd_amd import functional as Fn
    N = 2
    shapes = [(16, 20), (8, 10), (4, 5)]
    chans = [512, 1024, 2048]
    cs = [G.randn(10 + i, N, c, h, w).requires_grad_(True) for i, (c, (h, w)) in enumerate(zip(chans, shapes))]
    lw = [G.randn(20 + i, 256, c, 1, 1, scale=(1.0 / c) ** 0.5).requires_grad_(True) for i, c in enumerate(chans)]
    lb = [G.randn(30 + i, 256, scale=0.1).requires_grad_(True) for i in range(3)]
    fw = [G.randn(40 + i, 256, 256, 3, 3, scale=(1.0 / 2304) ** 0.5).requires_grad_(True) for i in range(5)]
    fb = [G.randn(50 + i, 256, scale=0.1).requires_grad_(True) for i in range(5)]
    lats = [F.conv2d(cs[i], lw[i], lb[i]) for i in range(3)]
    for i in range(2, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode="nearest")
    outs = [F.conv2d(lats[i], fw[i], fb[i], 1, 1) for i in range(3)]
    outs.append(F.conv2d(outs[-1], fw[3], fb[3], 2, 1))
    outs.append(F.conv2d(outs[-1], fw[4], fb[4], 2, 1))
    ref_cat = torch.cat([o.permute(0, 2, 3, 1).reshape(N, -1, 256) for o in outs], 1)
    dcat = G.randn(60, *ref_cat.shape)
    ref_cat.backward(dcat)
    csg = [c.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True) for c in cs]
    lwg = [cl_weight(w.detach()) for w in lw]
    lbg = [b.detach().cuda().requires_grad_(True) for b in lb]
    fwg = [cl_weight(w.detach()) for w in fw]
    fbg = [b.detach().cuda().requires_grad_(True) for b in fb]
    l = [Fn.ConvBias.apply(csg[i], lwg[i], lbg[i], 1, 1, 0) for i in range(3)]
    for i in range(2, 0, -1):
        l[i - 1] = Fn.UpsampleAdd.apply(l[i - 1], l[i])
    cat = Fn.FPNOutputs.apply(l[0], l[1], l[2], *fwg, *fbg)
    assert relerr(cat.detach().cpu(), ref_cat.detach()) < 2e-5
    g_in = dcat.cuda()
    g_keep = g_in.clone()
    cat.backward(g_in)
    assert torch.equal(g_in, g_keep)        # the caller's gradient tensor is not modified (the P5 / P6 slices accumulate into copies)
    for i in range(3):
        assert relerr(csg[i].grad.permute(0, 3, 1, 2).cpu(), cs[i].grad) < 5e-5, i
        assert relerr(lwg[i].grad.cpu(), lw[i].grad) < 5e-5, i
        assert relerr(lbg[i].grad.cpu(), lb[i].grad) < 5e-5, i
    for i in range(5):
        assert relerr(fwg[i].grad.cpu(), fw[i].grad) < 5e-5, i
        assert relerr(fbg[i].grad.cpu(), fb[i].grad) < 5e-5, i


def test_head_functions():
    from erd_amd import functional as Fn
    N = 2
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    A = sum(h * w for h, w in sizes)
    x = G.randn(70, N, A, 256).requires_grad_(True)
    w1 = G.randn(71, 256, 256, 3, 3, scale=(2.0 / 2304) ** 0.5).requires_grad_(True)
    g1, b1 = (0.5 + G.rand(72, 256)).requires_grad_(True), G.randn(73, 256, scale=0.2).requires_grad_(True)
    w2 = G.randn(74, 80, 256, 3, 3, scale=(1.0 / 2304) ** 0.5).requires_grad_(True)
    b2 = G.randn(75, 80, scale=0.1).requires_grad_(True)
    al = torch.tensor([0.9, 1.0, 1.1, 1.2, 1.3]).requires_grad_(True)
    outs = []
    off = 0
    for l, (h, w) in enumerate(sizes):
        xl = x[:, off:off + h * w].reshape(N, h, w, 256).permute(0, 3, 1, 2)
        y = F.relu(F.group_norm(F.conv2d(xl, w1, None, 1, 1), 32, g1, b1, 1e-5))
        o = F.conv2d(y, w2, b2, 1, 1) * al[l]
        outs.append(o.permute(0, 2, 3, 1).reshape(N, h * w, 80))
        off += h * w
    ref = torch.cat(outs, 1)
    dy = G.randn(76, *ref.shape)
    ref.backward(dy)
    xg = x.detach().cuda().requires_grad_(True)
    w1g, w2g = cl_weight(w1.detach()), cl_weight(w2.detach())
    g1g, b1g = g1.detach().cuda().requires_grad_(True), b1.detach().cuda().requires_grad_(True)
    b2g, alg = b2.detach().cuda().requires_grad_(True), al.detach().cuda().requires_grad_(True)
    y = Fn.HeadConvGN.apply(xg, w1g, g1g, b1g, sizes, 1e-5)
    pre = Fn.HeadConvBias.apply(y, w2g, b2g, sizes)
    out = Fn.LevelScale.apply(pre, alg, sizes)
    assert relerr(out.detach().cpu(), ref.detach()) < 2e-5
    out.backward(dy.cuda())
    assert relerr(xg.grad.cpu(), x.grad) < 1e-4
    assert relerr(w1g.grad.cpu(), w1.grad) < 1e-4
    assert relerr(g1g.grad.cpu(), g1.grad) < 2e-4 and relerr(b1g.grad.cpu(), b1.grad) < 2e-4
    assert relerr(w2g.grad.cpu(), w2.grad) < 1e-4 and relerr(b2g.grad.cpu(), b2.grad) < 1e-4
    assert relerr(alg.grad.cpu(), al.grad) < 1e-4


@pytest.mark.parametrize("cin,planes,stride,down", [(256, 64, 1, False), (256, 128, 2, True), (512, 128, 1, False),
                                                    (1024, 512, 2, True)])
def test_fused_bottleneck_block(cin, planes, stride, down):
    """the hand-scheduled bottleneck backward (ReLU masks / d-beta sums / shortcut add fused into GEMM epilogues)
    against torch autograd of the reference block (resnet.py:263-302)."""
    from erd_amd.modules import Bottleneck
    if not down:
        assert cin == planes * 4
    N, H, W = 2, 12, 14
    blk = Bottleneck(cin, planes, stride, down)
    sd = {}
    for i, (k, v) in enumerate(blk.state_dict().items()):
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[k] = v
        elif leaf == "running_var" or (leaf == "weight" and v.dim() == 1):
            sd[k] = 0.5 + G.rand(100 + i, *v.shape)
        elif v.dim() == 4:
            sd[k] = G.randn(100 + i, *v.shape, scale=(2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5)
        else:
            sd[k] = G.randn(100 + i, *v.shape, scale=0.2)
    blk.load_state_dict(sd)
    blk = blk.cuda()
    x = G.randn(90, N, cin, H, W).requires_grad_(True)
    ref = {k: v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v for k, v in sd.items()}

    def bn(t, p):
        return F.batch_norm(t, ref[p + ".running_mean"], ref[p + ".running_var"], ref[p + ".weight"], ref[p + ".bias"],
                            False, 0.0, 1e-5)
    o = F.relu(bn(F.conv2d(x, ref["conv1.weight"]), "bn1"))
    o = F.relu(bn(F.conv2d(o, ref["conv2.weight"], None, stride, 1), "bn2"))
    o = bn(F.conv2d(o, ref["conv3.weight"]), "bn3")
    idn = bn(F.conv2d(x, ref["downsample.0.weight"], None, stride), "downsample.1") if down else x
    y = F.relu(o + idn)
    dy = G.randn(91, *y.shape)
    y.backward(dy)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    out = blk(xg)
    assert relerr(out.detach().permute(0, 3, 1, 2).cpu(), y.detach()) < 2e-5
    out.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    assert relerr(xg.grad.permute(0, 3, 1, 2).cpu(), x.grad) < 1e-4
    for k, p in blk.named_parameters():
        assert relerr(p.grad.cpu(), ref[k].grad) < 3e-4, k


@pytest.mark.parametrize("sink", [False, True])
def test_res_layer_node_equals_the_chain_of_bottleneck_nodes(sink, monkeypatch):
    """ResLayerFn: a whole stage (projection block + identity blocks, res_layer.py:57-63) as one autograd node whose backward
    hands each block's dz3 straight to the block before it (the predecessor's ReLU mask and d-beta column sums ride in the
    epilogue of ONE input-gradient launch).  Same outputs bit for bit and the same gradients (float-atomic order only) as the
    chain of per-block nodes with their stand-alone ReLU backward passes -- with plain autograd accumulation and with the
    trainer's flat gradient slots (sink = True)."""
    from erd_amd import functional as Fn
    from erd_amd.modules import ResNet
    from erd_amd.engine import FlatParams

    def build():
        torch.manual_seed(5)
        net = ResNet(50, frozen_stages=1, norm_eval=True).cuda().train()
        with torch.no_grad():
            for m in net.modules():
                if hasattr(m, "running_var"):
                    m.running_var.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
        return net

    x = G.randn(7, 2, 3, 96, 128).cuda()
    res = {}
    for node in (False, True):
        monkeypatch.setattr(Fn, "RES_LAYER_NODE", node)
        net = build()
        flat = None
        if sink:
            named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
            named.reverse()
            flat = FlatParams(named, x.device)
            flat.zero_grad()
        outs = net(x)
        loss = sum((o.float() * G.randn(20 + i, *o.shape).cuda()).sum() for i, o in enumerate(outs))
        loss.backward()
        Fn.trail_join(x.device)
        torch.cuda.synchronize()
        res[node] = ([o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    assert res[False][1].keys() == res[True][1].keys() and len(res[True][1]) > 100
    for k, g in res[False][1].items():
        assert relerr(res[True][1][k].cpu(), g.cpu()) < 2e-5, k
