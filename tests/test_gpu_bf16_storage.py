"""GPU: BASELINE.json configs[2] with feature maps STORED as bf16 (erd_hip.h ERD_BF16; the reference's AMP switch,
tools/train.py:85-97, stores conv outputs in the low-precision type the same way).

Every kernel that reads or writes a map is checked against a PyTorch-CPU fp32 evaluation of the same op on the SAME
bf16-valued inputs.  The contract: inputs are widened exactly, the arithmetic is the fp32 arithmetic of the fp32-map
kernels, and the result is rounded ONCE (nearest even) when it is stored -- so the stored value is the bf16 neighbour of
the fp32 result: |hip - ref| <= 2^-8 |ref| + the fp32 kernel's own tolerance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G

ULP = 2.0 ** -8        # one bf16 rounding (round to nearest even: at most 2^-9 relative, 2^-8 leaves room for fp32 noise)


@pytest.fixture(scope="module")
def K():
    from erd_amd import kernels
    assert torch.cuda.is_available()
    return kernels


@pytest.fixture
def bf16_mode(K):
    K.set_compute("bf16")
    yield
    K.set_compute(K.DEFAULT_COMPUTE)


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc_b(t):      # NCHW cpu fp32 -> NHWC gpu bf16
    return t.permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)


def to_nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu()


def close_bf16(got, ref, extra=2e-5):
    """got is the bf16 neighbour of ref (one rounding) up to the fp32 kernel tolerance `extra` (relative to max |ref|)"""
    tol = ULP * ref.abs() + extra * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), f"{int(bad.sum())} of {bad.numel()} values off by more than one bf16 rounding; worst " \
                                f"{float(((got - ref).abs() - tol).max()):.3e}"


CASES = [
    # N, Cin, Cout, H, W, k, s, p
    (2, 64, 64, 20, 28, 1, 1, 0),
    (2, 64, 256, 20, 28, 1, 1, 0),
    (2, 128, 128, 26, 30, 3, 1, 1),
    (2, 128, 128, 26, 30, 3, 2, 1),
    (1, 256, 256, 25, 42, 3, 2, 1),
    (2, 256, 512, 20, 28, 1, 2, 0),
    (1, 2048, 512, 7, 11, 1, 1, 0),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CASES)
def test_conv_forward_bf16_maps(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    x = _r(G.randn(1, N, Cin, H, W))
    w = G.randn(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    scale = 0.5 + G.rand(3, Cout)
    shift = G.randn(4, Cout, scale=0.1)
    ref = F.conv2d(x, _r(w), None, s, p)
    OH, OW = ref.shape[2:]
    res = _r(G.randn(5, N, Cout, OH, OW))
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty((N, OH, OW, Cout), device="cuda", dtype=torch.bfloat16)
    K.conv_forward([nhwc_b(x)], wg, [out], k, s, p)
    close_bf16(to_nchw(out), ref)
    out2 = torch.empty_like(out)
    K.conv_forward([nhwc_b(x)], wg, [out2], k, s, p, scale=scale.cuda(), shift=shift.cuda(), res=[nhwc_b(res)], relu=True)
    close_bf16(to_nchw(out2), F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res))


def test_head_output_conv_bf16_in_fp32_out_and_back(K, bf16_mode):
    """gfl_cls / gfl_reg (gfl_head.py:224-229): bf16 tower features in, fp32 logits out (they feed the losses); backward:
    fp32 loss gradients in, bf16 feature gradients out; weight gradient from a bf16 map and an fp32 gradient"""
    N, Cin, Cout, H, W = 1, 256, 68, 13, 21
    x = _r(G.randn(41, N, Cin, H, W))
    w = G.randn(42, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5)
    b = G.randn(43, Cout, scale=0.1)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty((N, H, W, Cout), device="cuda")
    K.conv_forward([nhwc_b(x)], wg, [out], 3, 1, 1, shift=b.cuda())
    ref = F.conv2d(x, _r(w), b, 1, 1)
    assert float((to_nchw(out) - ref).abs().max() / ref.abs().max()) < 2e-5          # fp32 out: no rounding at all
    dy = G.randn(44, N, Cout, H, W)
    xv = x.clone().requires_grad_(True)
    wv = w.clone().requires_grad_(True)
    gx = torch.autograd.grad(F.conv2d(xv, _r(w), None, 1, 1), xv, _r(dy))[0]
    gw = torch.autograd.grad(F.conv2d(x, wv, None, 1, 1), wv, _r(dy))[0].permute(0, 2, 3, 1)
    dyg = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = torch.empty((N, H, W, Cin), device="cuda", dtype=torch.bfloat16)
    K.conv_dgrad([dyg], K.weight_transpose(wg, None), [dx], 3, 1, 1)
    close_bf16(to_nchw(dx), gx)
    part, S = K.conv_wgrad_partials([nhwc_b(x)], [dyg], 3, 1, 1)
    dW = torch.empty_like(wg)
    K.wgrad_reduce(part, S, wg, None, dW, False, None)
    assert float((dW.cpu() - gw).abs().max() / gw.abs().max()) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CASES)
def test_conv_dgrad_bf16_maps_with_mask_and_colsum(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(11, N, Cin, H, W).requires_grad_(True)
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    y = F.conv2d(x, _r(w), None, s, p)
    dy = _r(G.randn(13, *y.shape))
    xr = torch.autograd.grad(y, x, dy)[0]
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    wt = K.weight_transpose(wg, None)
    dx = torch.zeros((N, H, W, Cin), device="cuda", dtype=torch.bfloat16)
    K.conv_dgrad([nhwc_b(dy)], wt, [dx], k, s, p)
    close_bf16(to_nchw(dx), xr)
    if s == 1:      # fused ReLU mask of the producer + d beta column sums (BottleneckFn.backward)
        fwd = _r(G.randn(14, N, Cin, H, W))
        dx2 = torch.empty_like(dx)
        cs = torch.zeros(Cin, device="cuda")
        K.conv_dgrad([nhwc_b(dy)], wt, [dx2], k, s, p, relu_mask=[nhwc_b(fwd)], colsum=cs)
        refm = xr * (fwd > 0)
        close_bf16(to_nchw(dx2), refm)
        # the column sums are taken BEFORE the rounding (fp32 accumulators)
        assert float((cs.cpu() - refm.sum((0, 2, 3))).abs().max() / refm.sum((0, 2, 3)).abs().max()) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CASES)
def test_conv_wgrad_bf16_maps(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    x = _r(G.randn(11, N, Cin, H, W))
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    dy = _r(G.randn(13, *F.conv2d(x, w, None, s, p).shape))
    gw = torch.autograd.grad(F.conv2d(x, w, None, s, p), w, dy)[0].permute(0, 2, 3, 1)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    part, S = K.conv_wgrad_partials([nhwc_b(x)], [nhwc_b(dy)], k, s, p)
    dW = torch.empty_like(wg)
    K.wgrad_reduce(part, S, wg, None, dW, False, None)
    assert float((dW.cpu() - gw).abs().max() / gw.abs().max()) < 2e-5            # exact products, fp32 sums: no rounding


def test_relu_backward_and_colsum_bf16_maps(K, bf16_mode):
    N, H, W, Cc = 2, 13, 21, 256
    y = _r(G.randn(51, N, H, W, Cc))
    dy = _r(G.randn(52, N, H, W, Cc))
    dz, cs = K.relu_bwd_colsum(y.cuda().bfloat16(), dy.cuda().bfloat16(), True)
    ref = dy * (y > 0)
    assert dz.dtype == torch.bfloat16 and torch.equal(dz.float().cpu(), ref)      # a selection: exact
    assert float((cs.cpu() - ref.sum((0, 1, 2))).abs().max()) < 1e-3
    assert float((K.colsum(dy.cuda().bfloat16().reshape(-1, Cc)).cpu() - dy.sum((0, 1, 2))).abs().max()) < 1e-3


def test_group_norm_relu_bf16_maps(K, bf16_mode):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc, Gn = 2, 256, 32
    A = sum(h * w for h, w in sizes)
    c = _r(G.randn(61, N, A, Cc) * 1.5 + 0.3)
    gamma = 0.5 + G.rand(62, Cc)
    beta = G.randn(63, Cc, scale=0.2)
    dy = _r(G.randn(64, N, A, Cc))
    cv = c.clone().requires_grad_(True)
    gv, bv = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    outs, off = [], 0
    for (h, w) in sizes:
        lvl = cv[:, off:off + h * w].reshape(N, h * w, Cc).permute(0, 2, 1)
        outs.append(F.relu(F.group_norm(lvl, Gn, gv, bv, 1e-5)).permute(0, 2, 1))
        off += h * w
    ref = torch.cat(outs, 1)
    gc, gg, gb = torch.autograd.grad(ref, (cv, gv, bv), dy)
    cg = c.cuda().bfloat16()
    y, mr = K.gn_relu_forward(cg, gamma.cuda(), beta.cuda(), sizes, Gn, 1e-5)
    assert y.dtype == torch.bfloat16
    close_bf16(y.float().cpu(), ref.detach(), extra=1e-5)
    dc, dgamma, dbeta = K.gn_relu_backward(cg, dy.cuda().bfloat16(), gamma.cuda(), beta.cuda(), mr, sizes, Gn)
    assert dc.dtype == torch.bfloat16
    close_bf16(dc.float().cpu(), gc, extra=2e-5)
    assert float((dgamma.cpu() - gg).abs().max() / gg.abs().max()) < 1e-4         # fp32 reductions of exact inputs
    assert float((dbeta.cpu() - gb).abs().max() / gb.abs().max()) < 1e-4


def test_upsample_add_and_maxpool_bf16_maps(K, bf16_mode):
    N, Cc = 2, 256
    fine, coarse = _r(G.randn(71, N, 25, 42, Cc)), _r(G.randn(72, N, 13, 21, Cc))
    f = fine.cuda().bfloat16()
    K.upsample_add_(f, coarse.cuda().bfloat16())
    up = F.interpolate(coarse.permute(0, 3, 1, 2), size=(25, 42), mode="nearest").permute(0, 2, 3, 1)
    close_bf16(f.float().cpu(), fine + up, extra=0.0)
    dfine = _r(G.randn(73, N, 25, 42, Cc))
    dco = torch.zeros((N, 13, 21, Cc), device="cuda", dtype=torch.bfloat16)
    K.upsample_add_bwd_(dfine.cuda().bfloat16(), dco)
    cvar = coarse.clone().requires_grad_(True)
    g = torch.autograd.grad(F.interpolate(cvar.permute(0, 3, 1, 2), size=(25, 42), mode="nearest").permute(0, 2, 3, 1), cvar, dfine)[0]
    close_bf16(dco.float().cpu(), g, extra=1e-6)
    # stem: fp32 image -> bf16 pooled map
    x = G.randn(74, 1, 3, 64, 96)
    w = G.randn(75, 64, 3, 7, 7, scale=0.1)
    sc, sh = 0.5 + G.rand(76, 64), G.randn(77, 64, scale=0.1)
    z = K.stem(x.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), sc.cuda(), sh.cuda())
    assert z.dtype == torch.bfloat16
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, None, 2, 3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    close_bf16(to_nchw(z), ref, extra=2e-5)


def test_maps_of_one_role_must_share_a_storage_type(K, bf16_mode):
    x = torch.zeros((1, 8, 8, 64), device="cuda", dtype=torch.bfloat16)
    out = torch.zeros((1, 8, 8, 64), device="cuda", dtype=torch.bfloat16)
    res = torch.zeros((1, 8, 8, 64), device="cuda")
    w = torch.zeros((64, 1, 1, 64), device="cuda")
    with pytest.raises(AssertionError):
        K.conv_forward([x], w, [out], 1, 1, 0, res=[res])
    K.set_compute(K.DEFAULT_COMPUTE)
    with pytest.raises(AssertionError):          # bf16 maps only in the bf16 mode
        K.conv_forward([x], w, [out], 1, 1, 0)
