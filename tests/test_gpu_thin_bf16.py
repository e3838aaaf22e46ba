"""GPU: the bf16 matrix-core mode's form of the activation-stationary thin-K kernel (csrc/conv_thin.hip conv_thin_bf16_kernel; BASELINE
configs[2]: 1x1 convolutions with Cin in {64, 128} on bf16-STORED maps).  One MFMA per k16 step in the order of
conv_igemm_kernel<..., BF> without a K split: every form -- plain, folded BN + residual + ReLU, the input-gradient epilogue with
residual / ReLU mask / column sums, accumulate, stride 2, ragged last tile -- must be BIT-identical to that kernel
(erd_conv_thin_enable(0)); against an fp64 convolution of the bf16-rounded operands the distance is the output rounding's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G


@pytest.fixture()
def K():
    from erd_amd import kernels as K, _lib
    K.set_compute("bf16")
    lib = _lib.load()
    prev = lib.erd_conv_thin_enable(-1)
    yield K
    lib.erd_conv_thin_enable(prev)
    K.set_compute(K.DEFAULT_COMPUTE)


def both(fn):
    from erd_amd import _lib
    lib = _lib.load()
    out = []
    for on in (1, 0):
        lib.erd_conv_thin_enable(on)
        out.append(fn())
    lib.erd_conv_thin_enable(1)
    return out


def nhwc16(t):      # NCHW cpu fp32 -> NHWC gpu bf16
    return t.permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)


CASES = [(2, 128, 512, 25, 42, 1), (1, 64, 256, 30, 44, 1), (2, 128, 128, 13, 21, 1), (2, 64, 64, 17, 9, 1), (1, 128, 32, 40, 40, 1),
         (2, 128, 256, 26, 40, 2), (4, 128, 512, 100, 168, 1)]


@pytest.mark.parametrize("N,Cin,Cout,H,W,s", CASES)
def test_bf16_thin_forward_forms_are_bit_identical_to_the_igemm_kernel(K, N, Cin, Cout, H, W, s):
    x = G.randn(1, N, Cin, H, W)
    w = G.randn(2, Cout, Cin, 1, 1, scale=(2.0 / Cin) ** 0.5)
    scale, shift = 0.5 + G.rand(3, Cout), G.randn(4, Cout, scale=0.1)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    ref = F.conv2d(xb.double(), wb.double(), None, s, 0)
    OH, OW = ref.shape[2:]
    res = G.randn(5, N, Cout, OH, OW).to(torch.bfloat16)
    ref2 = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg, rg = nhwc16(x), nhwc16(res.float())

    def run():
        out = torch.empty((N, OH, OW, Cout), device="cuda", dtype=torch.bfloat16)
        K.conv_forward([xg], wg, [out], 1, s, 0)
        out2 = torch.empty_like(out)
        K.conv_forward([xg], wg, [out2], 1, s, 0, scale=scale.cuda(), shift=shift.cuda(), res=[rg], relu=True)
        return out, out2

    (a, a2), (b, b2) = both(run)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)) and torch.equal(a2.view(torch.int16), b2.view(torch.int16))
    e1 = float((a.float().permute(0, 3, 1, 2).cpu().double() - ref).norm() / ref.norm())
    e2 = float((a2.float().permute(0, 3, 1, 2).cpu().double() - ref2).norm() / ref2.norm())
    assert e1 < 4e-3 and e2 < 4e-3, (e1, e2)        # bf16 output rounding: 2^-9 per element


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 512, 128, 25, 42), (1, 256, 64, 30, 44), (4, 512, 128, 100, 168)])
def test_bf16_thin_input_gradient_forms(K, N, Cin, Cout, H, W):
    """the input gradient of a REDUCING 1x1 convolution (Cin -> Cout <= 128) is a thin GEMM K = Cout -> N = Cin with the fused
    bottleneck epilogue (shortcut gradient, ReLU mask, column sums)"""
    dz = G.randn(1, N, Cout, H, W)
    w = G.randn(2, Cout, Cin, 1, 1, scale=(2.0 / Cin) ** 0.5)
    short, mask = G.randn(4, N, Cin, H, W), G.randn(5, N, Cin, H, W)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    dzg, sg, mg = nhwc16(dz), nhwc16(short), nhwc16(mask)

    def run():
        wt = K.weight_transpose(wg)
        dx = torch.empty((N, H, W, Cin), device="cuda", dtype=torch.bfloat16)
        K.conv_dgrad([dzg], wt, [dx], 1, 1, 0)
        dx2 = torch.empty_like(dx)
        cs = torch.zeros((8, Cin), device="cuda")
        K.conv_dgrad([dzg], wt, [dx2], 1, 1, 0, res=[sg], relu_mask=[mg], colsum=cs)
        dx3 = sg.clone()
        K.conv_dgrad([dzg], wt, [dx3], 1, 1, 0, accumulate=True)
        return dx, dx2, cs.sum(0), dx3

    (a, a2, ca, a3), (b, b2, cb, b3) = both(run)
    for u, v in ((a, b), (a2, b2), (a3, b3)):
        assert torch.equal(u.view(torch.int16), v.view(torch.int16))
    assert torch.allclose(ca, cb, rtol=1e-5, atol=1e-3)
    ref = F.conv_transpose2d(dz.to(torch.bfloat16).double(), w.to(torch.bfloat16).double())
    assert float((a.float().permute(0, 3, 1, 2).cpu().double() - ref).norm() / ref.norm()) < 4e-3
