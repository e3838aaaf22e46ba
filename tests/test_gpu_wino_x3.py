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


@pytest.fixture()
def wino_p_mode():
    """ERD_WINO_P (read per launch by erd_wino_conv3x3_x3): 0 = items of 64 output channels (wino_x3_kernel), 2 = items of 128
    wherever Cout % 128 == 0 (wino_x3p_kernel), unset = the launch's own choice"""
    import os
    old = os.environ.get("ERD_WINO_P")

    def set_mode(m):
        if m is None:
            os.environ.pop("ERD_WINO_P", None)
        else:
            os.environ["ERD_WINO_P"] = str(m)
    yield set_mode
    set_mode(old)


P_SHAPES = [(2, 256, 256, [(26, 30)]), (1, 64, 128, [(20, 28)]), (2, 128, 128, [(25, 42)]), (1, 512, 512, [(7, 11)]), (1, 64, 256, [(34, 66)]),
            (2, 256, 256, [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]), (1, 128, 384, [(9, 12)]), (1, 256, 256, [(100, 168)])]


@pytest.mark.parametrize("N,Cin,Cout,sizes", P_SHAPES)
def test_wino_x3p_items_of_128_couts_equal_items_of_64_bit_for_bit(K, wino_p_mode, N, Cin, Cout, sizes):
    """wino_x3p_kernel (round 5: a tile block is transformed once per 128 output channels; eight waves that each multiply and
    transform; bulk output exchange) forms the same V values, the same MFMA sequence per accumulator and the same output sums as
    wino_x3_kernel: every epilogue form -- plain, folded BN + ReLU, residual aliasing the output + mask + fused column sums --, every
    block shape of the cover, several levels in one launch must come out BIT-identical (gfl_head.py:219-229, fpn.py:215,
    resnet.py:270-274 run on whichever the launch picks)."""
    from erd_amd.kernels import level_views
    A = sum(h * w for h, w in sizes)
    x = G.randn(71, N, A, Cin).cuda()
    w = G.randn(72, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5).permute(0, 2, 3, 1).contiguous().cuda()
    scale, shift = (0.5 + G.rand(73, Cout)).cuda(), G.randn(74, Cout, scale=0.1).cuda()
    base, mask = G.randn(75, N, A, Cout).cuda(), G.randn(76, N, A, Cout).cuda()
    U = K.wino_weights(w, x3=True)
    xs = level_views(x, sizes)
    got = {}
    for mode in (0, 2):
        wino_p_mode(mode)
        y1 = torch.full((N, A, Cout), float("nan"), device="cuda")
        K.wino_conv3x3(xs, U, level_views(y1, sizes), Cout)
        y2 = torch.full((N, A, Cout), float("nan"), device="cuda")
        K.wino_conv3x3(xs, U, level_views(y2, sizes), Cout, scale=scale, shift=shift, relu=True)
        y3 = base.clone()
        cs = torch.zeros(8, Cout, device="cuda")
        v3 = level_views(y3, sizes)
        K.wino_conv3x3(xs, U, v3, Cout, res=v3, mask=level_views(mask, sizes), colsum=cs)
        torch.cuda.synchronize()
        got[mode] = (y1, y2, y3, cs.sum(0))
    for a, b in zip(got[0][:3], got[2][:3]):
        assert not torch.isnan(b).any()
        assert torch.equal(a, b)
    # (the column sums are atomic adds into eight rows picked by workgroup index: same addends, another order)
    assert float((got[0][3] - got[2][3]).abs().max() / got[0][3].abs().max()) < 1e-5
    # ... and against fp64, once, so that "identical" cannot mean "identically wrong"
    off = 0
    for (h, ww) in sizes[:1]:
        ref = F.conv2d(x[:, off:off + h * ww].reshape(N, h, ww, Cin).permute(0, 3, 1, 2).double().cpu(),
                       w.permute(0, 3, 1, 2).double().cpu(), None, 1, 1)
        assert rel64(got[2][0][:, off:off + h * ww].reshape(N, h, ww, Cout).permute(0, 3, 1, 2).cpu(), ref) < 2e-6


def test_wino_x3_launch_picks_the_item_size_by_dispatch_rounds(K, wino_p_mode):
    """erd_wino_x3_couts_per_item (ABI v5) = the choice erd_wino_conv3x3_x3 makes: 128 output channels per item where Cout % 128 == 0
    and ceil(items128 / CUs) * 1.6 <= ceil(items64 / CUs) -- the head-tower launch (1 400 against 2 800 items) and a 100 x 168 FPN output,
    not a 50 x 84 map (280 against 560 items on 256 CUs), never Cout = 80."""
    from erd_amd import _lib
    from erd_amd.kernels import level_views, _fill_seg
    from erd_amd._lib import ConvSeg
    lib = _lib.load()

    def choice(N, sizes, Cout, Cin=256):
        A = sum(h * w for h, w in sizes)
        x, y = torch.empty(N, A, Cin, device="cuda"), torch.empty(N, A, Cout, device="cuda")
        xs, ys = level_views(x, sizes), level_views(y, sizes)
        segs = (ConvSeg * len(xs))()
        for i, (a, b) in enumerate(zip(xs, ys)):
            _fill_seg(segs[i], a, b, a.shape[1], a.shape[2], None, None, None)
        return int(lib.erd_wino_x3_couts_per_item(segs, len(xs), Cout))
    wino_p_mode(None)
    head = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    assert choice(4, head, 256) == 128
    assert choice(4, [(100, 168)], 256) == 128
    assert choice(4, [(50, 84)], 256) == 64
    assert choice(4, head, 80) == 64
    wino_p_mode(0)
    assert choice(4, head, 256) == 64
    wino_p_mode(2)
    assert choice(4, [(50, 84)], 256) == 128 and choice(4, head, 80) == 64
