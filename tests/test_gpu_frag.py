"""GPU: the fragment-streaming three-limb kernel for 1x1 convolutions with Cin >= 256 (csrc/conv_frag.hip; layer2-4 conv1 / conv3 /
shortcuts, the FPN laterals, and the input gradients with that GEMM shape: resnet.py:268-300, fpn.py:177-179).  Weight limb planes
pre-tiled as MFMA B-fragments (erd_weight_frag_x3) go global -> registers; the MFMA sequence per accumulator is the stream-K
kernel's, so with that kernel's K split off every form must be BIT-identical to it, and as close to fp64 as it is."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G
from test_gpu_kernels import nhwc, to_nchw


@pytest.fixture()
def K():
    from erd_amd import kernels as K, _lib
    K.set_compute("f32x3")
    lib = _lib.load()
    prev = lib.erd_conv_frag_enable(-1)
    keep = K.STREAMK
    K.STREAMK = False                  # (a stream-K split of the reference kernel changes its summation order)
    yield K
    K.STREAMK = keep
    lib.erd_conv_frag_enable(prev)
    K.set_compute(K.DEFAULT_COMPUTE)


def both(K, fn):
    from erd_amd import _lib
    lib = _lib.load()
    out = []
    for on in (1, 0):
        lib.erd_conv_frag_enable(on)
        K.FRAG = bool(on)
        out.append(fn())
    lib.erd_conv_frag_enable(1)
    K.FRAG = True
    return out


def test_weight_fragments_are_the_limb_planes_retiled(K):
    w = G.randn(1, 256, 512)
    wg = w.cuda()
    frag = K.weight_frag_x3(wg).view(3, 256 // 32, 512 // 16, 2, 32, 8).cpu()      # [plane][co / 32][k / 16][k half][co % 32][k % 8]
    planes = K.split3(wg).cpu()                                                    # [3, 256, 512]
    back = frag.permute(0, 1, 4, 2, 3, 5).reshape(3, 256, 512)
    assert torch.equal(back.view(torch.int16), planes.view(torch.int16))


CASES = [  # N, Cin, Cout, H, W, stride
    (2, 256, 128, 25, 42, 1), (1, 512, 256, 30, 44, 1), (2, 1024, 256, 13, 21, 1), (1, 256, 1024, 17, 9, 1),
    (2, 256, 512, 26, 40, 2), (4, 512, 128, 100, 168, 1), (4, 2048, 512, 25, 42, 1),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,s", CASES)
def test_frag_forward_forms_are_bit_identical_to_the_stream_k_kernel(K, N, Cin, Cout, H, W, s):
    x = G.randn(1, N, Cin, H, W)
    w = G.randn(2, Cout, Cin, 1, 1, scale=(2.0 / Cin) ** 0.5)
    scale, shift = 0.5 + G.rand(3, Cout), G.randn(4, Cout, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), None, s, 0)
    OH, OW = ref.shape[2:]
    res = G.randn(5, N, Cout, OH, OW)
    ref2 = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg, rg = nhwc(x), nhwc(res)

    def run():
        out = torch.empty((N, OH, OW, Cout), device="cuda")
        K.conv_forward([xg], wg, [out], 1, s, 0)
        out2 = torch.empty_like(out)
        K.conv_forward([xg], wg, [out2], 1, s, 0, scale=scale.cuda(), shift=shift.cuda(), res=[rg], relu=True)
        out3 = rg.clone()
        K.conv_forward([xg], wg, [out3], 1, s, 0, res=[out3])
        return out, out2, out3

    (a, a2, a3), (b, b2, b3) = both(K, run)
    assert torch.equal(a, b) and torch.equal(a2, b2) and torch.equal(a3, b3)
    e1 = float((to_nchw(a).double() - ref).norm() / ref.norm())
    e2 = float((to_nchw(a2).double() - ref2).norm() / ref2.norm())
    print("frag %d->%d on %dx%d/%d: rel L2 to fp64 plain %.2e, epilogue %.2e" % (Cin, Cout, H, W, s, e1, e2))
    assert e1 < 5e-7 and e2 < 5e-7


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 128, 256, 25, 42), (1, 256, 512, 30, 44), (4, 256, 1024, 50, 84)])
def test_frag_input_gradient_forms(K, N, Cin, Cout, H, W):
    """the input gradient of an EXPANDING 1x1 convolution (Cin -> Cout >= 256) is a GEMM K = Cout -> N = Cin with the fused
    bottleneck epilogue (shortcut gradient, ReLU mask, column sums)"""
    dz = G.randn(1, N, Cout, H, W)
    w = G.randn(2, Cout, Cin, 1, 1, scale=(2.0 / Cin) ** 0.5)
    rowscale = 0.5 + G.rand(3, Cout)
    short, mask = G.randn(4, N, Cin, H, W), G.randn(5, N, Cin, H, W)
    ref = F.conv_transpose2d(dz.double() * rowscale.double().view(1, -1, 1, 1), w.double())
    ref_m = (ref + short.double()) * (mask.double() > 0)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    dzg, sg, mg = nhwc(dz), nhwc(short), nhwc(mask)

    def run():
        wt = K.weight_transpose(wg, rowscale.cuda())
        dx = torch.empty((N, H, W, Cin), device="cuda")
        K.conv_dgrad([dzg], wt, [dx], 1, 1, 0)
        dx2 = torch.empty_like(dx)
        cs = torch.zeros((8, Cin), device="cuda")
        K.conv_dgrad([dzg], wt, [dx2], 1, 1, 0, res=[sg], relu_mask=[mg], colsum=cs)
        dx3 = sg.clone()
        K.conv_dgrad([dzg], wt, [dx3], 1, 1, 0, accumulate=True)
        return dx, dx2, cs.sum(0), dx3

    (a, a2, ca, a3), (b, b2, cb, b3) = both(K, run)
    assert torch.equal(a, b) and torch.equal(a2, b2) and torch.equal(a3, b3)
    assert torch.allclose(ca, cb, rtol=1e-5, atol=1e-4)
    assert float((to_nchw(a).double() - ref).norm() / ref.norm()) < 3e-7
    assert float((to_nchw(a2).double() - ref_m).norm() / ref_m.norm()) < 3e-7
    assert torch.allclose(ca.cpu().double(), ref_m.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
