"""GPU: every C-ABI kernel group against a plain PyTorch-CPU fp32 reference of the same op
(conv / norm / pooling) -- the per-anchor ERD kernels are checked against the oracle and the
golden fixtures in test_gpu_losses.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import golden_inputs as G


@pytest.fixture(scope="module")
def K():
    from erd_amd import kernels
    assert torch.cuda.is_available()
    return kernels


def dev(t):
    return t.cuda()


def nhwc(t):      # NCHW cpu -> NHWC gpu contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def to_nchw(t):   # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).cpu()


def relerr(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


CONV_CASES = [
    # N, Cin, Cout, H, W, k, s, p
    (2, 64, 64, 20, 28, 1, 1, 0),
    (2, 64, 256, 20, 28, 1, 1, 0),
    (1, 256, 128, 25, 42, 1, 1, 0),
    (2, 128, 128, 26, 30, 3, 1, 1),
    (2, 128, 128, 26, 30, 3, 2, 1),
    (1, 256, 256, 13, 21, 3, 2, 1),     # odd sizes (P6-like 25x42 -> 13x21 is below)
    (1, 256, 256, 25, 42, 3, 2, 1),
    (2, 256, 512, 20, 28, 1, 2, 0),     # downsample
    (1, 256, 80, 13, 21, 3, 1, 1),      # gfl_cls
    (1, 256, 68, 7, 11, 3, 1, 1),       # gfl_reg
    (1, 2048, 512, 7, 11, 1, 1, 0),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CONV_CASES)
def test_conv_forward_epilogues(K, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(1, N, Cin, H, W)
    w = G.randn(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    scale = 0.5 + G.rand(3, Cout)
    shift = G.randn(4, Cout, scale=0.1)
    ref = F.conv2d(x, w, None, s, p)
    OH, OW = ref.shape[2:]
    res = G.randn(5, N, Cout, OH, OW)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg = nhwc(x)
    # plain
    out = torch.empty((N, OH, OW, Cout), device="cuda")
    K.conv_forward([xg], wg, [out], k, s, p)
    assert relerr(to_nchw(out), ref) < 2e-5
    # folded BN + residual + relu
    out2 = torch.empty_like(out)
    K.conv_forward([xg], wg, [out2], k, s, p, scale=scale.cuda(), shift=shift.cuda(), res=[nhwc(res)], relu=True)
    ref2 = F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    assert relerr(to_nchw(out2), ref2) < 2e-5
    # bias + per-level scalar
    alpha = torch.tensor(1.37)
    out3 = torch.empty_like(out)
    K.conv_forward([xg], wg, [out3], k, s, p, shift=shift.cuda(), alphas=[alpha.cuda()])
    assert relerr(to_nchw(out3), (ref + shift.view(1, -1, 1, 1)) * alpha) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CONV_CASES)
def test_conv_dgrad_wgrad(K, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(11, N, Cin, H, W).requires_grad_(True)
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = G.randn(13, *y.shape)
    y.backward(dy)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    rowscale = 0.5 + G.rand(14, Cout)
    # dgrad (with rowscale folded into the transposed weights == dy*scale)
    wt = K.weight_transpose(wg, rowscale.cuda())
    dx = torch.zeros((N, H, W, Cin), device="cuda")
    K.conv_dgrad([nhwc(dy)], wt, [dx], k, s, p)
    xr = torch.autograd.grad(F.conv2d(x, w, None, s, p), x, dy * rowscale.view(1, -1, 1, 1))[0]
    assert relerr(to_nchw(dx), xr) < 2e-5
    # accumulate form
    base = G.randn(15, N, Cin, H, W)
    dx2 = nhwc(base)
    wt1 = K.weight_transpose(wg, None)
    K.conv_dgrad([nhwc(dy)], wt1, [dx2], k, s, p, accumulate=True)
    assert relerr(to_nchw(dx2), x.grad + base) < 2e-5
    # wgrad + reduce (+ rowdot)
    part, S = K.conv_wgrad_partials([nhwc(x.detach())], [nhwc(dy)], k, s, p)
    dW = torch.empty_like(wg)
    rowdot = torch.empty(Cout, device="cuda")
    K.wgrad_reduce(part, S, wg, rowscale.cuda(), dW, False, rowdot)
    gw = w.grad.permute(0, 2, 3, 1)
    assert relerr(dW.cpu(), gw * rowscale.view(-1, 1, 1, 1)) < 5e-5
    assert relerr(rowdot.cpu(), (gw * w.detach().permute(0, 2, 3, 1)).sum((1, 2, 3))) < 5e-4


def test_conv_multilevel_shared_weights(K):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc = 2, 256
    A = sum(h * w for h, w in sizes)
    x = G.randn(21, N, A, Cc)
    w = G.randn(22, Cc, Cc, 3, 3, scale=(2.0 / (Cc * 9)) ** 0.5)
    xg = x.cuda()
    out = torch.empty((N, A, Cc), device="cuda")
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    K.conv_forward(K.level_views(xg, sizes), wg, K.level_views(out, sizes), 3, 1, 1)
    off = 0
    for (h, wd) in sizes:
        xl = x[:, off:off + h * wd].reshape(N, h, wd, Cc).permute(0, 3, 1, 2)
        ref = F.conv2d(xl, w, None, 1, 1).permute(0, 2, 3, 1).reshape(N, h * wd, Cc)
        assert relerr(out[:, off:off + h * wd].cpu(), ref) < 2e-5
        off += h * wd


def test_stem_and_maxpool(K):
    x = G.randn(31, 2, 3, 67, 93)
    w = G.randn(32, 64, 3, 7, 7, scale=0.1)
    scale, shift = 0.5 + G.rand(33, 64), G.randn(34, 64, scale=0.1)
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1)
    out = K.stem(x.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), scale.cuda(), shift.cuda())
    assert relerr(to_nchw(out), ref) < 2e-5


def test_bn_fold_and_backward_pieces(K):
    Cc = 128
    g, b, m, v = 0.5 + G.rand(41, Cc), G.randn(42, Cc), G.randn(43, Cc), 0.5 + G.rand(44, Cc)
    sc, sh = K.bn_fold(g.cuda(), b.cuda(), m.cuda(), v.cuda())
    s_ref = g / torch.sqrt(v + 1e-5)
    assert relerr(sc.cpu(), s_ref) < 1e-6 and relerr(sh.cpu(), b - m * s_ref) < 1e-6
    y = G.randn(45, 2, 9, 11, Cc)
    dy = G.randn(46, 2, 9, 11, Cc)
    dz, cs = K.relu_bwd_colsum(y.cuda(), dy.cuda(), True)
    ref = dy * (y > 0)
    assert torch.equal(dz.cpu(), ref)
    assert relerr(cs.cpu(), ref.sum((0, 1, 2))) < 1e-5
    rd, db = G.randn(47, Cc), G.randn(48, Cc)
    dg = K.bn_dgamma(rd.cuda(), db.cuda(), m.cuda(), v.cuda())
    assert relerr(dg.cpu(), (rd - m * db) / torch.sqrt(v + 1e-5)) < 1e-5


@pytest.mark.parametrize("copies,Cc", [(1, 64), (8, 2048), (16, 72), (64, 128), (128, 64), (32, 1000)])
def test_bn_dgamma_folds_replicated_column_sums(K, copies, Cc):
    """the [copies, C] accumulators of the input-gradient epilogues (functional.py rep_slot): d gamma AND the folded d beta, for every
    replication the step uses, a ragged channel count included (C % 32 != 0: the fold kernel's last workgroup)"""
    m, v = G.randn(61, Cc), 0.5 + G.rand(62, Cc)
    rd, rep = G.randn(63, Cc), G.randn(64, copies, Cc)
    fold = torch.full((Cc,), float("nan"), device="cuda")
    dg = K.bn_dgamma(rd.cuda(), rep.cuda(), m.cuda(), v.cuda(), dbeta_out=fold)
    db = rep.double().sum(0)
    assert relerr(fold.cpu().double(), db) < 1e-6
    assert relerr(dg.cpu().double(), (rd.double() - m.double() * db) / torch.sqrt(v.double() + 1e-5)) < 1e-5
    if copies <= 8:          # the sum order of the serial loop the kernel replaced: rows in order
        seq = rep[0].clone()
        for r in range(1, copies):
            seq += rep[r]
        assert torch.equal(fold.cpu(), seq)


def test_groupnorm_relu_fwd_bwd(K):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc = 2, 256
    A = sum(h * w for h, w in sizes)
    c = G.randn(51, N, A, Cc, scale=2.0, shift=0.3)
    gamma, beta = 0.5 + G.rand(52, Cc), G.randn(53, Cc, scale=0.3)
    dy = G.randn(54, N, A, Cc)
    y, mr = K.gn_relu_forward(c.cuda(), gamma.cuda(), beta.cuda(), sizes)
    dc, dg, db = K.gn_relu_backward(c.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), mr, sizes)
    off = 0
    dg_ref, db_ref = torch.zeros(Cc), torch.zeros(Cc)
    for (h, w) in sizes:
        cl = c[:, off:off + h * w].reshape(N, h, w, Cc).permute(0, 3, 1, 2).clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        yl = F.relu(F.group_norm(cl, 32, gr, br, 1e-5))
        yl.backward(dy[:, off:off + h * w].reshape(N, h, w, Cc).permute(0, 3, 1, 2))
        flat = lambda t: t.permute(0, 2, 3, 1).reshape(N, h * w, Cc)
        assert relerr(y[:, off:off + h * w].cpu(), flat(yl.detach())) < 1e-5
        assert relerr(dc[:, off:off + h * w].cpu(), flat(cl.grad)) < 1e-4
        dg_ref += gr.grad
        db_ref += br.grad
        off += h * w
    assert relerr(dg.cpu(), dg_ref) < 1e-4 and relerr(db.cpu(), db_ref) < 1e-4


@pytest.mark.parametrize("sizes,N", [([(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)], 2), ([(25, 38), (13, 19), (7, 10), (4, 5), (2, 3)], 3)])
def test_tower_layer_with_groupnorm_statistics_from_the_producer(K, sizes, N):
    """gfl_head.py:158-177 as ONE fused unit (kernels.conv3x3_gn_relu_forward): the three-limb Winograd kernel's output stage accumulates
    the (sum, sum of squares) of every item's 16 groups, a tiny launch folds them to (mean, 1 / std) -- erd_wino_conv3x3_x3_gn -- and
    erd_gn_relu_apply normalises.  Against the three-pass form (conv, statistics pass, finalize, apply) on the same input: the convolution output bit for bit,
    mean / rstd and the normalised output to 1e-6 (another summation order of the same fp32 values, f64 accumulation on both sides),
    and the statistics against an fp64 evaluation of the convolution output itself."""
    Cc = 256
    A = sum(h * w for h, w in sizes)
    x = G.randn(81, N, A, Cc).cuda()
    w = (G.randn(82, Cc, 3, 3, Cc) * (2.0 / (9 * Cc)) ** 0.5).cuda()
    gamma, beta = (0.5 + G.rand(83, Cc)).cuda(), G.randn(84, Cc, scale=0.3).cuda()
    keep = K.GN_FUSED
    try:
        K.GN_FUSED = True
        c1, y1, mr1 = K.conv3x3_gn_relu_forward(x, w, gamma, beta, sizes)
        K.GN_FUSED = False
        c0, y0, mr0 = K.conv3x3_gn_relu_forward(x, w, gamma, beta, sizes)
    finally:
        K.GN_FUSED = keep
    torch.cuda.synchronize()
    assert torch.equal(c1, c0)
    assert relerr(mr1.cpu(), mr0.cpu()) < 1e-6, relerr(mr1.cpu(), mr0.cpu())
    assert float((y1 - y0).abs().max()) < 1e-5 * float(y0.abs().max())
    off = 0
    cd = c0.double().cpu()
    for l, (h, ww) in enumerate(sizes):
        blk = cd[:, off:off + h * ww].reshape(N, h * ww, 32, 8)
        mean = blk.mean(dim=(1, 3))
        var = blk.var(dim=(1, 3), unbiased=False)
        assert relerr(mr1[:, l, :, 0].double().cpu(), mean) < 1e-5
        assert relerr(mr1[:, l, :, 1].double().cpu(), 1.0 / torch.sqrt(var + 1e-5)) < 1e-5
        off += h * ww


def test_cu_reserve_resizes_the_grids_and_keeps_the_results(K):
    """erd_set_cu_reserve (ERDTrainer.tune_cu_reserve under data parallelism): the persistent Winograd grid, the stream-K grid and the
    one-round weight-gradient split are sized for (CUs - reserve).  Results: the tile-parallel and Winograd launches bit for bit (the
    items are the same, only who runs them changes), stream-K and split-K launches to fp32 rounding (another partition of the K axis)."""
    from erd_amd import _lib
    lib = _lib.load()
    full = int(lib.erd_usable_cus())
    x = G.randn(91, 2, 50, 84, 256).cuda()
    w3 = (G.randn(92, 256, 3, 3, 256) * 0.02).cuda()
    w1 = (G.randn(93, 256, 1, 1, 1024) * 0.03).cuda()
    x1 = G.randn(94, 2, 50, 84, 1024).cuda()
    dy = G.randn(95, 2, 50, 84, 256).cuda()

    def run():
        y3 = torch.empty(2, 50, 84, 256, device="cuda")
        K.conv_forward([x], w3, [y3], 3, 1, 1)                      # Winograd, persistent grid
        y1 = torch.empty(2, 50, 84, 256, device="cuda")
        K.conv_forward([x1], w1, [y1], 1, 1, 0)                     # K = 1024: stream-K
        part, S = K.conv_wgrad_partials([x], [dy], 3, 1, 1)
        dW = torch.empty_like(w3)
        K.wgrad_reduce(part, S, w3, None, dW, False, None)
        torch.cuda.synchronize()
        return y3, y1, dW, S
    try:
        a = run()
        assert K.set_cu_reserve(8) == 0 and int(lib.erd_usable_cus()) == full - 8
        b = run()
    finally:
        K.set_cu_reserve(0)
    assert int(lib.erd_usable_cus()) == full
    assert torch.equal(a[0], b[0])
    assert relerr(b[1].cpu(), a[1].cpu()) < 1e-6 and relerr(b[2].cpu(), a[2].cpu()) < 1e-6
    assert b[3] <= a[3]                                              # fewer CUs: no more split-K siblings than before


def test_upsample_add_and_adjoint(K):
    N, Cc = 2, 256
    fine, coarse = G.randn(61, N, 10, 14, Cc), G.randn(62, N, 5, 7, Cc)
    f = fine.cuda()
    K.upsample_add_(f, coarse.cuda())
    ref = fine.permute(0, 3, 1, 2) + F.interpolate(coarse.permute(0, 3, 1, 2), size=(10, 14), mode="nearest")
    assert torch.allclose(to_nchw(f), ref)
    dc = torch.zeros((N, 5, 7, Cc), device="cuda")
    K.upsample_add_bwd_(fine.cuda(), dc)
    cr = coarse.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.interpolate(cr, size=(10, 14), mode="nearest").backward(fine.permute(0, 3, 1, 2))
    assert relerr(to_nchw(dc), cr.grad) < 1e-6


def test_level_scale_colsum_sgd(K):
    sizes = [(6, 7), (3, 4), (2, 2)]
    A = sum(h * w for h, w in sizes)
    x, dy = G.randn(71, 2, A, 68), G.randn(72, 2, A, 68)
    al = torch.tensor([0.9, 1.1, 1.3])
    y = K.level_scale(x.cuda(), al.cuda(), sizes)
    dx, dal = K.level_scale_bwd(x.cuda(), dy.cuda(), al.cuda(), sizes)
    off = 0
    for i, (h, w) in enumerate(sizes):
        sl = slice(off, off + h * w)
        assert torch.allclose(y[:, sl].cpu(), x[:, sl] * al[i])
        assert torch.allclose(dx[:, sl].cpu(), dy[:, sl] * al[i])
        assert float(dal[i]) == pytest.approx(float((dy[:, sl] * x[:, sl]).sum()), rel=1e-4)
        off += h * w
    assert relerr(K.colsum(x.cuda()).cpu(), x.sum((0, 1))) < 1e-5
    n = 4096 + 8
    p, g = G.randn(73, n), G.randn(74, n)
    pg, buf = p.cuda(), torch.zeros(n, device="cuda")
    ref_p = [p.clone()]
    opt = torch.optim.SGD(ref_p, lr=0.02, momentum=0.9, weight_decay=1e-4)
    for it in range(3):
        ref_p[0].grad = g.clone()
        opt.step()
        K.sgd_momentum_(pg, g.cuda(), buf, 0.02, 0.9, 1e-4, 1.0, it == 0)
    assert relerr(pg.cpu(), ref_p[0]) < 1e-6


def test_data_preprocessor_bit_exact_vs_oracle():
    """DetDataPreprocessor (R21): BGR->RGB, (x-mean)/std, zero pad to /32 after normalisation -- bit-exact vs the oracle
    on ragged uint8 images; float inputs and a non-zero pad value too."""
    from erd_amd import MODELS, DetDataSample
    from oracle import erd_oracle as O
    rng = np.random.RandomState(11)
    imgs = [torch.from_numpy(rng.randint(0, 256, size=(3, h, w), dtype=np.uint8)) for h, w in [(123, 153), (97, 160), (128, 31)]]
    want, metas = O.preprocess(imgs)
    pre = MODELS.build(dict(type="DetDataPreprocessor", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375],
                            bgr_to_rgb=True, pad_size_divisor=32)).cuda()
    samples = [DetDataSample() for _ in imgs]
    out = pre(dict(inputs=imgs, data_samples=samples), True)
    assert out["inputs"].shape == want.shape and torch.equal(out["inputs"].cpu(), want)
    for s, m in zip(samples, metas):
        assert s.metainfo["img_shape"] == m["img_shape"] and s.metainfo["pad_shape"] == m["pad_shape"] \
            and s.metainfo["batch_input_shape"] == m["batch_input_shape"]
    # float inputs, no channel flip, pad value 7
    pre2 = MODELS.build(dict(type="DetDataPreprocessor", mean=[1.5, 2.5, 3.5], std=[2.0, 3.0, 7.0], pad_size_divisor=16,
                             pad_value=7)).cuda()
    f = [torch.from_numpy(rng.rand(3, 20, 33).astype(np.float32)) * 255]
    got = pre2(dict(inputs=f), False)["inputs"].cpu()
    ref = torch.full((1, 3, 32, 48), 7.0)
    ref[0, :, :20, :33] = (f[0] - torch.tensor([1.5, 2.5, 3.5]).view(3, 1, 1)) / torch.tensor([2.0, 3.0, 7.0]).view(3, 1, 1)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("Cout", [70, 10, 2, 129])
def test_conv_forward_cout_not_multiple_of_4(K, Cout):
    """a 70-class teacher head (BASELINE configs[3]): output rows are not 16-B aligned -> the scalar epilogue"""
    N, Cin, H, W = 2, 256, 13, 21
    x = G.randn(31, N, Cin, H, W)
    w = G.randn(32, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5)
    b = G.randn(33, Cout, scale=0.1)
    ref = F.conv2d(x, w, b, 1, 1)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    K.conv_forward([nhwc(x)], w.permute(0, 2, 3, 1).contiguous().cuda(), [out], 3, 1, 1, shift=b.cuda())
    assert relerr(to_nchw(out), ref) < 2e-5
    res = G.randn(34, N, Cout, H, W)
    out2 = torch.empty_like(out)
    K.conv_forward([nhwc(x)], w.permute(0, 2, 3, 1).contiguous().cuda(), [out2], 3, 1, 1, scale=(0.5 + G.rand(35, Cout)).cuda(),
                   shift=b.cuda(), res=[nhwc(res)], relu=True)
    assert relerr(to_nchw(out2), F.relu(F.conv2d(x, w, None, 1, 1) * (0.5 + G.rand(35, Cout)).view(1, -1, 1, 1) +
                                        b.view(1, -1, 1, 1) + res)) < 2e-5


@pytest.fixture
def bf16_mode(K):
    K.set_compute("bf16")
    yield
    K.set_compute(K.DEFAULT_COMPUTE)


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", CONV_CASES + [(1, 256, 70, 13, 21, 3, 1, 1)])
def test_conv_forward_bf16_matrix_cores(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    """BASELINE configs[2]: multiplicands rounded to bf16 (exact products), fp32 accumulate/epilogue/output -- must agree
    with conv(bf16(x), bf16(w)) evaluated in fp32 as tightly as the fp32 path agrees with conv(x, w)."""
    x = G.randn(1, N, Cin, H, W)
    w = G.randn(2, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    scale = 0.5 + G.rand(3, Cout)
    shift = G.randn(4, Cout, scale=0.1)
    ref = F.conv2d(_r(x), _r(w), None, s, p)
    OH, OW = ref.shape[2:]
    res = G.randn(5, N, Cout, OH, OW)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty((N, OH, OW, Cout), device="cuda")
    K.conv_forward([nhwc(x)], wg, [out], k, s, p)
    assert relerr(to_nchw(out), ref) < 2e-5
    assert relerr(to_nchw(out), F.conv2d(x, w, None, s, p)) > 1e-4        # it really is the bf16 product
    out2 = torch.empty_like(out)
    K.conv_forward([nhwc(x)], wg, [out2], k, s, p, scale=scale.cuda(), shift=shift.cuda(), res=[nhwc(res)], relu=True)
    assert relerr(to_nchw(out2), F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", [c for c in CONV_CASES if c[2] % 4 == 0])
def test_conv_dgrad_bf16_matrix_cores(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(11, N, Cin, H, W).requires_grad_(True)
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5)
    y = F.conv2d(x, _r(w), None, s, p)
    dy = G.randn(13, *y.shape)
    xr = torch.autograd.grad(y, x, _r(dy))[0]                 # conv_transpose(bf16(dy), bf16(w)) in fp32
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    wt = K.weight_transpose(wg, None)
    dx = torch.zeros((N, H, W, Cin), device="cuda")
    K.conv_dgrad([nhwc(dy)], wt, [dx], k, s, p)
    assert relerr(to_nchw(dx), xr) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p", [c for c in CONV_CASES if c[2] % 4 == 0])
def test_conv_wgrad_bf16_matrix_cores(K, bf16_mode, N, Cin, Cout, H, W, k, s, p):
    x = G.randn(11, N, Cin, H, W)
    w = G.randn(12, Cout, Cin, k, k, scale=(2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    dy = G.randn(13, *F.conv2d(x, w, None, s, p).shape)
    gw = torch.autograd.grad(F.conv2d(_r(x), w, None, s, p), w, _r(dy))[0].permute(0, 2, 3, 1)
    wg = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    part, S = K.conv_wgrad_partials([nhwc(x)], [nhwc(dy)], k, s, p)
    dW = torch.empty_like(wg)
    K.wgrad_reduce(part, S, wg, None, dW, False, None)
    assert relerr(dW.cpu(), gw) < 2e-5


def test_conv_wgrad_bf16_multilevel(K, bf16_mode):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc, Co = 2, 256, 80
    A = sum(h * w for h, w in sizes)
    x = G.randn(21, N, A, Cc)
    dz = G.randn(23, N, A, Co)
    xs, dzs, ref, off = [], [], 0, 0
    xg, dg = x.cuda(), dz.cuda()
    for (h, w) in sizes:
        xl = x[:, off:off + h * w].reshape(N, h, w, Cc).permute(0, 3, 1, 2)
        dl = dz[:, off:off + h * w].reshape(N, h, w, Co).permute(0, 3, 1, 2)
        wz = torch.zeros(Co, Cc, 3, 3, requires_grad=True)
        ref = ref + torch.autograd.grad(F.conv2d(_r(xl), wz, None, 1, 1), wz, _r(dl))[0]
        xs.append(xg[:, off:off + h * w].unflatten(1, (h, w)))
        dzs.append(dg[:, off:off + h * w].unflatten(1, (h, w)))
        off += h * w
    part, S = K.conv_wgrad_partials(xs, dzs, 3, 1, 1)
    dW = torch.empty((Co, 3, 3, Cc), device="cuda")
    K.wgrad_reduce(part, S, dW, None, dW, False, None)
    assert relerr(dW.cpu(), ref.permute(0, 2, 3, 1)) < 2e-5


@pytest.mark.parametrize("Cin,Co", [(256, 256), (256, 80), (64, 68), (320, 136), (128, 40), (256, 64)])
def test_conv_wgrad_fp32_multilevel_three_tap_kernel(K, Cin, Co, monkeypatch):
    """3x3 / stride 1 weight gradient over the head's five concatenated levels: the three-taps-per-workgroup kernel
    (16-pixel row chunks with halo; widths 28 / 14 / 7 / 4 / 2 exercise partial chunks and rows narrower than a chunk)
    against autograd, and against the generic one-tap kernel (ERD_WGRAD_ROW3=0)."""
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N = 2
    A = sum(h * w for h, w in sizes)
    x = G.randn(21, N, A, Cin)
    dz = G.randn(23, N, A, Co)
    xs, dzs, ref, off = [], [], 0, 0
    xg, dg = x.cuda(), dz.cuda()
    for (h, w) in sizes:
        xl = x[:, off:off + h * w].reshape(N, h, w, Cin).permute(0, 3, 1, 2)
        dl = dz[:, off:off + h * w].reshape(N, h, w, Co).permute(0, 3, 1, 2)
        wz = torch.zeros(Co, Cin, 3, 3, requires_grad=True)
        ref = ref + torch.autograd.grad(F.conv2d(xl, wz, None, 1, 1), wz, dl)[0]
        xs.append(xg[:, off:off + h * w].unflatten(1, (h, w)))
        dzs.append(dg[:, off:off + h * w].unflatten(1, (h, w)))
        off += h * w
    outs = []
    for row3 in ("1", "0"):
        monkeypatch.setenv("ERD_WGRAD_ROW3", row3)
        part, S = K.conv_wgrad_partials(xs, dzs, 3, 1, 1)
        dW = torch.empty((Co, 3, 3, Cin), device="cuda")
        K.wgrad_reduce(part, S, dW, None, dW, False, None)
        assert relerr(dW.cpu(), ref.permute(0, 2, 3, 1)) < 2e-5, row3
        outs.append(dW.cpu())
    assert relerr(outs[0], outs[1]) < 5e-6


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 256, 256, 26, 30), (1, 64, 64, 20, 28), (2, 128, 128, 25, 42), (1, 256, 80, 13, 21),
                                            (1, 512, 512, 7, 11), (1, 256, 68, 8, 16), (1, 64, 32, 5, 3),
                                            # ragged tile grids: every block shape of the cover (1x32 / 2x16 bottom strips,
                                            # 32x1 / 16x2 / 8x4 right strips)
                                            (1, 64, 64, 34, 66), (2, 64, 64, 12, 20), (1, 64, 96, 100, 168), (1, 128, 64, 50, 84)])
def test_winograd_conv3x3_forward(K, N, Cin, Cout, H, W):
    """Winograd F(2x2,3x3) == direct 3x3 stride-1 convolution (odd sizes: partial tiles; Cout not a multiple of 64)"""
    x = G.randn(41, N, Cin, H, W)
    w = G.randn(42, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5)
    scale = 0.5 + G.rand(43, Cout)
    shift = G.randn(44, Cout, scale=0.1)
    ref = F.conv2d(x, w, None, 1, 1)
    U = K.wino_weights(w.permute(0, 2, 3, 1).contiguous().cuda())
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    K.wino_conv3x3([nhwc(x)], U, [out], Cout)
    assert relerr(to_nchw(out), ref) < 5e-5
    out2 = torch.empty_like(out)
    K.wino_conv3x3([nhwc(x)], U, [out2], Cout, scale=scale.cuda(), shift=shift.cuda(), relu=True)
    assert relerr(to_nchw(out2), F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))) < 5e-5


def test_winograd_conv3x3_multilevel_and_input_gradient(K):
    sizes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    N, Cc = 2, 256
    A = sum(h * w for h, w in sizes)
    x = G.randn(51, N, A, Cc)
    w = G.randn(52, Cc, Cc, 3, 3, scale=(2.0 / (Cc * 9)) ** 0.5)
    wg = w.permute(0, 2, 3, 1).contiguous().cuda()
    xg = x.cuda()
    out = torch.empty((N, A, Cc), device="cuda")
    from erd_amd.kernels import level_views
    K.wino_conv3x3(level_views(xg, sizes), K.wino_weights(wg), level_views(out, sizes), Cc)
    off = 0
    for (h, ww) in sizes:
        xl = x[:, off:off + h * ww].reshape(N, h, ww, Cc).permute(0, 3, 1, 2)
        ref = F.conv2d(xl, w, None, 1, 1)
        got = out[:, off:off + h * ww].reshape(N, h, ww, Cc).permute(0, 3, 1, 2).cpu()
        assert relerr(got, ref) < 5e-5, (h, ww)
        off += h * ww
    # input gradient = the same kernel on dz with the flipped, transposed weights
    xr = G.randn(53, 1, Cc, 13, 21).requires_grad_(True)
    dy = G.randn(54, 1, Cc, 13, 21)
    gx = torch.autograd.grad(F.conv2d(xr, w, None, 1, 1), xr, dy)[0]
    wt = torch.empty((Cc, 3, 3, Cc), device="cuda")
    from erd_amd._lib import call
    call("erd_weight_transpose", wg.data_ptr(), None, wt.data_ptr(), Cc, 9, Cc, 1, torch.cuda.current_stream().cuda_stream)
    dx = torch.empty((1, 13, 21, Cc), device="cuda")
    K.wino_conv3x3([nhwc(dy)], K.wino_weights(wt), [dx], Cc)
    assert relerr(to_nchw(dx), gx) < 5e-5


@pytest.mark.parametrize("H,W", [(13, 21), (50, 84), (25, 42), (7, 11), (100, 168), (27, 33)])
def test_winograd_cover_stores_every_pixel_exactly_once(K, H, W):
    """The block-shape regions of the Winograd cover (interior 4x8 blocks, flat bottom strips, narrow right strips) must
    PARTITION the map: a tall right-strip block reaches below the interior rows, and before the regions carried their own
    limits those tiles were stored twice -- invisible for a plain store, but a residual that aliases the output
    (accumulate form) was added twice and the fused column sums counted the pixels twice (13x21, 50x84 and 27x33 overlap;
    25x42, 7x11, 100x168 do not)."""
    N, Cin, Cout = 1, 64, 64
    x = G.randn(81, N, Cin, H, W)
    w = G.randn(82, Cout, Cin, 3, 3, scale=(2.0 / (Cin * 9)) ** 0.5)
    ref = F.conv2d(x, w, None, 1, 1)
    U = K.wino_weights(w.permute(0, 2, 3, 1).contiguous().cuda())
    base = G.randn(83, N, Cout, H, W)
    for _ in range(3):          # (the double store was a race: repeat)
        out = nhwc(base)
        cs = torch.zeros(Cout, device="cuda")
        K.wino_conv3x3([nhwc(x)], U, [out], Cout, res=[out], colsum=cs)
        assert relerr(to_nchw(out), ref + base) < 2e-5
        assert relerr(cs.cpu(), (ref + base).sum((0, 2, 3))) < 1e-4
