"""GPU: the "f32x3" compute mode (erd_conv_desc::w_x3) -- fp32 maps, fp32 accumulation, fp32 results, with every product
formed on the bf16 matrix cores from exact three-limb splits of both fp32 multiplicands (six of the nine limb products; what
is dropped is below 2^-23 of the product).  gfx950's fp32 MFMA runs at 1/16 of the bf16 rate, so this is how an fp32-accurate
GEMM is done fast on this chip.  Checked here: the limb split is exact; every direct implicit-GEMM form (forward with its
epilogues, input gradient incl. the merged stride-2 classes, multi-level launches, stream-K splits) is AS CLOSE TO AN FP64
evaluation as the native fp32-MFMA kernel is; the prepared limb planes of a trainer equal the per-use ones bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G
from test_gpu_kernels import CONV_CASES, nhwc, to_nchw


@pytest.fixture()
def K():
    from erd_amd import kernels as K
    yield K
    K.set_compute(K.DEFAULT_COMPUTE)


def rel64(a, b):
    """relative L2 distance of a (fp32, cpu) from the fp64 reference b"""
    return float((a.double() - b).norm() / (b.norm() + 1e-300))


def test_split3_is_exact(K):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1 << 16, generator=g) * torch.exp(8 * torch.randn(1 << 16, generator=g))      # 20 orders of magnitude
    x[:4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-30])
    planes = K.split3(x.cuda()).cpu()
    assert planes.dtype == torch.bfloat16 and planes.shape == (3, 1 << 16)
    hi, mid, lo = planes[0].float(), planes[1].float(), planes[2].float()
    assert torch.equal((hi + mid) + lo, x)                     # the three limbs ARE the value (8 + 8 + 8 significand bits)
    assert torch.equal(hi.view(torch.int32) & 0xffff, torch.zeros_like(hi, dtype=torch.int32))
    # round-to-nearest limbs: each remainder is at most half an ulp of the limb above, and sign-symmetric (a truncating split
    # leaves remainders of the value's own sign, twice as large: the limb products a three-limb GEMM drops were then a bias)
    assert bool((mid.abs() <= hi.abs() * 2.0 ** -8).all()) and bool((lo.abs() <= hi.abs() * 2.0 ** -16).all())
    assert torch.equal(hi, x.bfloat16().float())                # plane 0 IS the value's bf16 neighbour
    nz = (x != 0) & (mid != 0)
    assert 0.45 < float((torch.sign(mid[nz]) == torch.sign(x[nz])).float().mean()) < 0.55


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CONV_CASES)
def test_f32x3_convolutions_are_as_close_to_fp64_as_the_native_fp32_kernels(K, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(1, N, Cin, H, W)
    w = G.randn(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    scale, shift = 0.5 + G.rand(3, Cout), G.randn(4, Cout, scale=0.1)
    ref = F.conv2d(x.double(), w.double(), None, s, p)
    OH, OW = ref.shape[2:]
    res = G.randn(5, N, Cout, OH, OW)
    ref2 = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + res.double())
    dy = G.randn(13, N, Cout, OH, OW)
    rowscale = 0.5 + G.rand(14, Cout)
    xd = x.double().requires_grad_(True)
    gref = torch.autograd.grad(F.conv2d(xd, w.double(), None, s, p), xd, dy.double() * rowscale.double().view(1, -1, 1, 1))[0]
    wd = w.double().requires_grad_(True)
    wref = torch.autograd.grad(F.conv2d(x.double(), wd, None, s, p), wd, dy.double())[0].permute(0, 2, 3, 1)      # [Cout, k, k, Cin]
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg, dyg = nhwc(x), nhwc(dy)
    err = {}
    for mode in ("f32", "f32x3"):
        K.set_compute(mode)
        keep, K.WINOGRAD = K.WINOGRAD, False          # (the direct kernels are what the mode changes)
        try:
            out = torch.empty((N, OH, OW, Cout), device="cuda")
            K.conv_forward([xg], wg, [out], k, s, p)
            out2 = torch.empty_like(out)
            K.conv_forward([xg], wg, [out2], k, s, p, scale=scale.cuda(), shift=shift.cuda(), res=[nhwc(res)], relu=True)
            wt = K.weight_transpose(wg, rowscale.cuda())
            dx = torch.zeros((N, H, W, Cin), device="cuda")
            K.conv_dgrad([dyg], wt, [dx], k, s, p)
            part, S = K.conv_wgrad_partials([xg], [dyg], k, s, p)
            dW = torch.empty_like(wg)
            K.wgrad_reduce(part, S, wg, None, dW, False, None)
        finally:
            K.WINOGRAD = keep
        err[mode] = (rel64(to_nchw(out), ref), rel64(to_nchw(out2), ref2), rel64(to_nchw(dx), gref), rel64(dW.cpu(), wref))
    print("rel L2 to fp64 (plain / epilogue / input gradient / weight gradient): native fp32 MFMA %.2e %.2e %.2e %.2e | f32x3 %.2e %.2e %.2e %.2e"
          % (err["f32"] + err["f32x3"]))
    for a, b in zip(err["f32x3"], err["f32"]):
        assert a <= max(1.5 * b, 2e-7), err


def test_f32x3_multilevel_and_streamk_launches(K):
    """five levels sharing one weight in one launch, and a long-K launch that the stream-K decomposition splits"""
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc = 2, 256
    A = sum(h * w for h, w in sizes)
    x = G.randn(21, N, A, Cc)
    w = G.randn(22, Cc, Cc, 3, 3, scale=(2.0 / (Cc * 9)) ** 0.5)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    K.set_compute("f32x3")
    keep, K.WINOGRAD = K.WINOGRAD, False
    try:
        out = torch.empty((N, A, Cc), device="cuda")
        K.conv_forward(K.level_views(x.cuda(), sizes), wg, K.level_views(out, sizes), 3, 1, 1)
        off = 0
        for (h, wd) in sizes:
            xl = x[:, off:off + h * wd].reshape(N, h, wd, Cc).permute(0, 3, 1, 2)
            ref = F.conv2d(xl.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(N, h * wd, Cc)
            assert rel64(out[:, off:off + h * wd].cpu(), ref) < 3e-7
            off += h * wd
        # 1x1 on 2048 channels, 4 x 25 x 42 pixels: 33 x 4 tiles of 64 K-slices on 512 resident workgroups -> split K
        x2 = G.randn(23, 4, 2048, 25, 42)
        w2 = G.randn(24, 512, 2048, 1, 1, scale=(2.0 / 2048) ** 0.5)
        ref = F.conv2d(x2.double(), w2.double())
        o2 = torch.empty((4, 25, 42, 512), device="cuda")
        for _ in range(2):      # twice: the tickets of the first launch must be back at zero
            K.conv_forward([nhwc(x2)], w2.permute(0, 2, 3, 1).contiguous().cuda(), [o2], 1, 1, 0)
            assert rel64(to_nchw(o2), ref) < 6e-7       # (K = 2048: the native kernel reads 3.4e-7 here)
    finally:
        K.WINOGRAD = keep


def test_f32x3_operand_slices_by_lds_dma_equal_register_staging_bit_for_bit(K, monkeypatch):
    """Round 6: the three-limb implicit GEMM stages its K-slices by LDS-DMA (`buffer_load ... lds`, ERD_IG_GLDS=1, the default) instead of
    global -> registers -> ds_write (ERD_IG_GLDS=0).  Same LDS image (the swizzle moves to the source address, padding taps and
    channels past Cin arrive as zeros from out-of-range buffer offsets), same MFMA sequence: every form bit for bit -- a 3x3 / stride-2
    convolution with padding on a ragged map, five levels in one launch (direct kernels), a stream-K launch (K = 2048), Cin = 68 (a
    slice that ends inside a weight chunk), and a masked accumulating input gradient with column sums."""
    K.set_compute("f32x3")
    keep, K.WINOGRAD = K.WINOGRAD, False
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    A = sum(h * w for h, w in sizes)
    x5 = G.randn(31, 2, A, 256).cuda()
    w5 = (G.randn(32, 256, 3, 3, 256) * 0.02).cuda()
    xs2 = G.randn(33, 2, 37, 53, 128).cuda()
    ws2 = (G.randn(34, 256, 3, 3, 128) * 0.03).cuda()
    xk = G.randn(35, 4, 25, 42, 2048).cuda()
    wk = (G.randn(36, 512, 1, 1, 2048) * 0.02).cuda()
    x68 = G.randn(37, 2, 13, 21, 68).cuda()
    w68 = (G.randn(38, 256, 3, 3, 68) * 0.04).cuda()
    dy = G.randn(39, 2, 37, 53, 512).cuda()
    wd = (G.randn(40, 512, 1, 1, 128) * 0.05).cuda()
    mask, base = G.randn(41, 2, 37, 53, 128).cuda(), G.randn(42, 2, 37, 53, 128).cuda()
    sc, sh = (0.5 + G.rand(43, 256)).cuda(), G.randn(44, 256, scale=0.1).cuda()

    def run():
        o5 = torch.empty((2, A, 256), device="cuda")
        K.conv_forward(K.level_views(x5, sizes), w5, K.level_views(o5, sizes), 3, 1, 1, scale=sc, shift=sh, relu=True)
        o2 = torch.empty((2, 19, 27, 256), device="cuda")
        K.conv_forward([xs2], ws2, [o2], 3, 2, 1)
        ok = torch.empty((4, 25, 42, 512), device="cuda")
        K.conv_forward([xk], wk, [ok], 1, 1, 0)
        o68 = torch.empty((2, 13, 21, 256), device="cuda")
        K.conv_forward([x68], w68, [o68], 3, 1, 1)
        acc = base.clone()
        cs = torch.zeros(8, 128, device="cuda")
        K.conv_dgrad([dy], K.weight_transpose(wd), [acc], 1, 1, 0, accumulate=True, relu_mask=[mask], colsum=cs)
        torch.cuda.synchronize()
        return o5, o2, ok, o68, acc
    try:
        monkeypatch.setenv("ERD_IG_GLDS", "0")
        ref = run()
        monkeypatch.setenv("ERD_IG_GLDS", "1")
        got = run()
    finally:
        K.WINOGRAD = keep
    for i, (a, b) in enumerate(zip(ref, got)):
        assert not torch.isnan(b).any() and torch.equal(a, b), i


@pytest.mark.parametrize("Cin,Cout,sizes", [(256, 256, [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]), (128, 128, [(26, 30)]),
                                             (256, 80, [(13, 21), (7, 11)]), (256, 68, [(13, 21)]), (512, 512, [(9, 17)]),
                                             (64, 64, [(20, 36)])])
def test_f32x3_three_tap_weight_gradient_is_as_close_to_fp64_as_the_fp32_kernel(K, Cin, Cout, sizes):
    """the three-limb form of the 3x3 / stride-1 weight gradient (both operands split in the loader, kernel-row taps by register
    shifts of the packed pixel vectors) against an fp64 evaluation, next to the fp32-MFMA three-tap kernel: ragged widths (not
    multiples of 16), several levels summed in one launch, 80 / 68 / 64 output channels"""
    N = 2
    A = sum(h * w for h, w in sizes)
    x = G.randn(41, N, A, Cin)
    dz = G.randn(42, N, A, Cout)
    ref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64)
    off = 0
    for (h, w) in sizes:
        xl = x[:, off:off + h * w].reshape(N, h, w, Cin).permute(0, 3, 1, 2).double()
        dl = dz[:, off:off + h * w].reshape(N, h, w, Cout).permute(0, 3, 1, 2).double()
        ref += torch.nn.grad.conv2d_weight(xl, (Cout, Cin, 3, 3), dl, stride=1, padding=1)
        off += h * w
    ref = ref.permute(0, 2, 3, 1)                  # [Cout, 3, 3, Cin]
    xg, dg = x.cuda(), dz.cuda()
    err = {}
    for mode in ("f32", "f32x3"):
        K.set_compute(mode)
        part, S = K.conv_wgrad_partials(K.level_views(xg, sizes), K.level_views(dg, sizes), 3, 1, 1)
        dW = torch.empty((Cout, 3, 3, Cin), device="cuda")
        K.wgrad_reduce(part, S, dW, None, dW, False, None)
        err[mode] = rel64(dW.cpu(), ref)
    print("weight gradient %d -> %d, rel L2 to fp64: fp32 MFMA %.2e | three-limb %.2e" % (Cin, Cout, err["f32"], err["f32x3"]))
    assert err["f32x3"] <= max(1.5 * err["f32"], 3e-7), err
