"""GPU: Winograd F(2x2,3x3) in the three-limb form (winograd.hip wino_x3_kernel, erd_wino_conv3x3_x3): the 16 transform-domain GEMMs
on the bf16 matrix cores from exact limb splits of U = G g G^T and V = B^T d B, fp32 maps / accumulation / results.  Every form the
fp32 Winograd kernel serves -- plain, folded-BN + ReLU, residual + column sums (accumulate), ReLU mask, several levels in one
launch, ragged maps with every block shape of the cover, Cout not a multiple of 32 / 4, the input-gradient form -- must be as close
to an fp64 convolution as the fp32 Winograd kernel is (the gfl_head.py:219-229 towers, fpn.py:215 outputs and resnet.py:270-274
conv2 layers run on it in the default "f32x3" mode)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G
from test_gpu_kernels import nhwc, to_nchw


@pytest.fixture()
def K():
    from erd_amd import kernels as K
    yield K
    K.set_compute(K.DEFAULT_COMPUTE)


def rel64(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-300))


SHAPES = [(2, 256, 256, 26, 30), (1, 64, 64, 20, 28), (2, 128, 128, 25, 42), (1, 256, 80, 13, 21), (1, 512, 512, 7, 11),
          (1, 256, 68, 8, 16), (1, 64, 32, 5, 3), (1, 256, 70, 9, 12),
          (1, 64, 64, 34, 66), (2, 64, 64, 12, 20), (1, 64, 96, 100, 168), (1, 128, 64, 50, 84)]


@pytest.mark.parametrize("N,Cin,Cout,H,W", SHAPES)
def test_wino_x3_forward_forms_against_fp64(K, N, Cin, Cout, H, W):
    x = G.randn(41, N, Cin, H, W)
    w = G.randn(42, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5)
    scale, shift = 0.5 + G.rand(43, Cout), G.randn(44, Cout, scale=0.1)
    base, mask = G.randn(45, N, Cout, H, W), G.randn(46, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    ref2 = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    ref3 = (ref + base.double()) * (mask.double() > 0)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg = nhwc(x)
    err = {}
    for x3 in (False, True):
        U = K.wino_weights(wg, x3=x3)
        assert (U.dtype == torch.bfloat16) == x3
        out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
        K.wino_conv3x3([xg], U, [out], Cout)
        out2 = torch.empty_like(out)
        K.wino_conv3x3([xg], U, [out2], Cout, scale=scale.cuda(), shift=shift.cuda(), relu=True)
        e3 = cs_err = 0.0
        if Cout % 4 == 0:
            for _ in range(2):          # residual aliasing the output + mask + fused column sums: every pixel exactly once
                out3 = nhwc(base)
                cs = torch.zeros(Cout, device="cuda")
                K.wino_conv3x3([xg], U, [out3], Cout, res=[out3], mask=[nhwc(mask)], colsum=cs)
                e3 = max(e3, rel64(to_nchw(out3), ref3))
                cs_err = max(cs_err, rel64(cs.cpu(), ref3.sum((0, 2, 3))))
        err[x3] = (rel64(to_nchw(out), ref), rel64(to_nchw(out2), ref2), e3, cs_err)
    print("%d->%d %dx%d rel L2 to fp64 (plain / BN+ReLU / res+mask / colsum): fp32 Winograd %.2e %.2e %.2e %.2e | three-limb %.2e %.2e %.2e %.2e"
          % ((Cin, Cout, H, W) + err[False] + err[True]))
    for a, b in zip(err[True][:3], err[False][:3]):
        assert a <= max(1.5 * b, 2e-6), err
    assert err[True][3] < 1e-4


def test_wino_x3_head_levels_and_input_gradient(K):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc = 2, 256
    A = sum(h * w for h, w in sizes)
    x = G.randn(51, N, A, Cc)
    w = G.randn(52, Cc, Cc, 3, 3, scale=(2.0 / (Cc * 9)) ** 0.5)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg = x.cuda()
    from erd_amd.kernels import level_views
    out = torch.empty((N, A, Cc), device="cuda")
    K.wino_conv3x3(level_views(xg, sizes), K.wino_weights(wg, x3=True), level_views(out, sizes), Cc)
    off = 0
    for (h, ww) in sizes:
        xl = x[:, off:off + h * ww].reshape(N, h, ww, Cc).permute(0, 3, 1, 2)
        ref = F.conv2d(xl.double(), w.double(), None, 1, 1)
        got = out[:, off:off + h * ww].reshape(N, h, ww, Cc).permute(0, 3, 1, 2).cpu()
        assert rel64(got, ref) < 2e-6, (h, ww)
        off += h * ww
    # input gradient = the same kernel on dz with the flipped, transposed weights
    xr = G.randn(53, 1, Cc, 13, 21).double().requires_grad_(True)
    dy = G.randn(54, 1, Cc, 13, 21)
    gx = torch.autograd.grad(F.conv2d(xr, w.double(), None, 1, 1), xr, dy.double())[0]
    wt = K.weight_transpose(wg)
    dx = torch.empty((1, 13, 21, Cc), device="cuda")
    K.wino_conv3x3([nhwc(dy)], K.wino_weights(wt, flip=True, x3=True), [dx], Cc)
    assert rel64(to_nchw(dx), gx) < 2e-6


def test_wino_x3_full_size_head_tower_launch_runs_twice_identically(K):
    """the head-tower launch of the benchmark (4 x 22 400 pixels, 256 -> 256, five levels): the persistent grid walks ~3 000 items
    with the pair exchange through LDS flags at every item switch -- two runs must agree bit for bit, and with the fp32 Winograd
    kernel to fp32 rounding"""
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    N, Cc = 4, 256
    A = sum(h * w for h, w in sizes)
    from erd_amd.kernels import level_views
    xg = G.randn(61, N, A, Cc).cuda()
    wg = G.randn(62, Cc, Cc, 3, 3, scale=(2.0 / (Cc * 9)) ** 0.5).permute(0, 2, 3, 1).contiguous().cuda()
    outs = []
    for x3 in (True, True, False):
        out = torch.empty((N, A, Cc), device="cuda")
        K.wino_conv3x3(level_views(xg, sizes), K.wino_weights(wg, x3=x3), level_views(out, sizes), Cc)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[2]).norm() / outs[2].norm()) < 2e-6
