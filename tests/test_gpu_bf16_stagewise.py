"""GPU: the bf16 mode (BASELINE configs[2]; DESIGN 8.0c: bf16 matrix cores, bf16-STORED maps) against the oracle's `bf16_stored_maps` mode,
STAGE BY STAGE with the oracle's own stage inputs ("teacher forcing"), forward and backward, at 800x1333.

Why not end to end (VERDICT r4 "parity soft spot 1"; tools/dbg/bf16_layer_probe.py, bf16_parity_probe.py): rounding every stored map to
bf16 makes the network chaotic at the 4e-3 level.  Two implementations that round at the SAME places still sum in different orders; a
1e-7 difference in a pre-rounding value lands on the other side of a bf16 rounding boundary for ~0.05 % of the elements per store, each
such flip is a full bf16 ulp (4e-3), feeds ~1e-4 of noise into everything downstream, and the flips multiply: against the oracle with
identical rounding points the HIP path's stage outputs differ in 0.6 % of the elements after stem + layer1, 32 % after layer2, 56 %
after layer3 -- from layer3 on it is as far from that oracle (8e-3) as bf16 is from fp32, the ERS sets overlap 0.92-0.94 either way and
the end-to-end gradient cosine to ANY second implementation sits at the bf16-vs-fp32 noise floor (0.989-0.991).  An end-to-end bound
therefore cannot be tighter than that floor, and a 1 % defect in one kernel hides under it (measured: a BN scale x 1.01 moves the
end-to-end cosine from 0.99077 to 0.98812).

What CAN be tight: one UNIT at a time, both sides fed the SAME bf16 inputs ("teacher forcing").  Inside a unit only its own two to four
stores flip, and the result sits at the rounding level of one bf16 map -- so that the same 1 % defect stands out (negative control
below).  The backward pass needs the finer units: through a whole stage of six bottlenecks even the ORACLE's bf16 modes are 14 % from its
fp32 backward (ReLU masks of flipped activations), and the HIP path 9 % from the oracle's.  Units (reference code in brackets): stem +
layer1 forward [resnet.py:631-640]; every trainable Bottleneck of layer2-4, forward + input gradient + parameter gradients
[resnet.py:263-302]; the FPN as a whole [fpn.py:161-221]; every conv3x3 -> GroupNorm -> ReLU layer of both towers on all five levels
[gfl_head.py:158-177, 219-222] and the two output convolutions with their Scales [gfl_head.py:224-229] -- each against torch autograd of
the oracle's unit under the same upstream gradient; whole-stage FORWARD results as well."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from e2e_util import build_erd, f7_state_dicts
from oracle import erd_oracle as O

# measured at 800x1333, seeds 7 / 8 (printed by the test): bottleneck forward 0.7-3.0e-4, tower layer 4-5e-5, output convolutions 1.6e-7
# (fp32 results); input gradients 1.2e-4 - 5.8e-3, parameter gradients 1.6e-3 - 6.5e-3 (the oracle rounds a gradient map where autograd
# hands it over, the HIP path where its kernels store one: dz behind the ReLU mask, the residual sum in two steps -- rounding noise of a
# bf16 map, 2.3e-3, a few times over); FPN forward 1.3e-3 (five chained stores); whole stages forward 1.9e-3 / 5.0e-3 / 2.7e-3.
FWD_TOL = 1e-3          # relative L2 of a unit's output
FPN_FWD_TOL = 3e-3
STAGE_FWD_TOL = 8e-3    # a whole stage forward (up to six bottlenecks of flips)
BWD_TOL = 9e-3          # relative L2 of a unit's input gradient / of all its parameter gradients together
DEFECT = 1.01           # negative control: one BN scale x 1.01 must break FWD_TOL by a wide margin


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def to_map(t, grad=False):
    """oracle NCHW fp32 (bf16-representable) -> the HIP path's NHWC map in its storage type"""
    from erd_amd import kernels as K
    m = t.detach().permute(0, 2, 3, 1).contiguous().to(K.act_dtype()).cuda()
    return m.requires_grad_(True) if grad else m


def from_map(m):
    return m.detach().float().permute(0, 3, 1, 2).cpu()


def bf16_noise(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(shape, generator=g)).to(torch.bfloat16).to(torch.float32)


@pytest.fixture(scope="module")
def setup():
    from erd_amd import kernels as K
    tsd, ssd = f7_state_dicts()
    K.set_compute("bf16")
    model = build_erd(tsd, ssd)
    yield ssd, model
    K.set_compute(K.DEFAULT_COMPUTE)


def _oracle_chain(ssd, seed):
    """the oracle's stored-map forward, stage by stage: image, C2..C5, P3..P7"""
    imgs, _, _ = O.synthetic_batch(1, 800, 1333, 40, seed=seed)
    x, _ = O.preprocess(imgs)
    sub = {k[len("backbone."):]: v for k, v in ssd.items() if k.startswith("backbone.")}
    with torch.no_grad(), O.bf16_stored_maps():
        c = [O.resnet_layer(sub, O.resnet_stem(sub, x), 0)]
        for li in (1, 2, 3):
            c.append(O.resnet_layer(sub, c[-1], li))
        p = O.fpn_forward(ssd, c)
    return x, sub, c, p


def _param_grads(model, prefix):
    return {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if k.startswith(prefix) and p.grad is not None}


def _pg_err(got, ref):
    assert set(ref) <= set(got), sorted(set(ref) - set(got))[:4]
    num = sum(float((got[k].double() - ref[k].double()).pow(2).sum()) for k in ref)
    den = sum(float(ref[k].double().pow(2).sum()) for k in ref)
    return (num / den) ** 0.5


@pytest.mark.parametrize("seed", [7, 8])
def test_bf16_mode_unit_by_unit_against_the_oracle_with_stored_maps(setup, seed):
    from erd_amd import functional as Fn, kernels as K
    ssd, model = setup
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    try:
        x, sub, c, p = _oracle_chain(ssd, seed)
        rows = []
        # ---- stem + layer1 (frozen: forward only), and the whole trainable stages forward
        with torch.no_grad():
            h, _ = model.backbone.trunk(x.cuda())
            rows.append(("stem+layer1", rel(from_map(h), c[0]), None, None, FWD_TOL))
            for li in (1, 2, 3):
                hh = to_map(c[li - 1])
                for blk in getattr(model.backbone, f"layer{li + 1}"):
                    hh = blk(hh)
                rows.append((f"layer{li + 1} (whole)", rel(from_map(hh), c[li]), None, None, STAGE_FWD_TOL))
        # ---- every trainable bottleneck: forward, input gradient, parameter gradients
        for li in (1, 2, 3):
            xin_chain = c[li - 1]
            for b, blk in enumerate(getattr(model.backbone, f"layer{li + 1}")):
                name = f"layer{li + 1}.{b}"
                leaf = {k: (v.clone().requires_grad_(True) if k.startswith(name + ".") and v.dtype == torch.float32 and "running" not in k else v)
                        for k, v in sub.items()}
                xin = xin_chain.clone().requires_grad_(True)
                with O.bf16_stored_maps():
                    y = O.resnet_block(leaf, xin, li, b)
                    g = bf16_noise(y.shape, 1000 * li + b, float(y.detach().std()))
                    y.backward(g)
                model.zero_grad(set_to_none=True)
                hin = to_map(xin_chain, grad=True)
                hh = blk(hin)
                hh.backward(to_map(g))
                ref = {("backbone." + k): v.grad for k, v in leaf.items() if v.requires_grad and v.grad is not None}
                rows.append((name, rel(from_map(hh), y.detach()), rel(from_map(hin.grad), xin.grad),
                             _pg_err(_param_grads(model, f"backbone.{name}."), ref), FWD_TOL))
                xin_chain = y.detach()
        # ---- FPN
        gs = [bf16_noise(v.shape, 200 + i, float(v.std())) for i, v in enumerate(p)]
        leaf = {k: (v.clone().requires_grad_(True) if k.startswith("neck.") else v) for k, v in ssd.items()}
        cin = [v.clone().requires_grad_(True) for v in c]
        with O.bf16_stored_maps():
            po = O.fpn_forward(leaf, cin)
            torch.autograd.backward(po, gs)
        model.zero_grad(set_to_none=True)
        hin = [to_map(v, grad=(i > 0)) for i, v in enumerate(c)]
        cat, sizes = model.neck.forward_cat([m.permute(0, 3, 1, 2) for m in hin])
        cat.backward(torch.cat([to_map(g_).reshape(1, -1, 256) for g_ in gs], 1))
        fwd = max(rel(from_map(v), po[i].detach()) for i, v in enumerate(K.level_views(cat, sizes)))
        dxs = max(rel(from_map(hin[i].grad), cin[i].grad) for i in (1, 2, 3))
        rows.append(("fpn", fwd, dxs, _pg_err(_param_grads(model, "neck."), {k: v.grad for k, v in leaf.items() if k.startswith("neck.")}), FPN_FWD_TOL))
        # ---- the towers, one conv -> GN -> ReLU layer at a time on all five levels
        sizes = [tuple(v.shape[-2:]) for v in p]
        flat = lambda maps: torch.cat([v.permute(0, 2, 3, 1).reshape(1, -1, v.shape[1]) for v in maps], 1)
        for branch, convs in (("cls", model.bbox_head.cls_convs), ("reg", model.bbox_head.reg_convs)):
            chain = [v for v in p]
            for i, m in enumerate(convs):
                pre = f"bbox_head.{branch}_convs.{i}."
                leaf = {k: (v.clone().requires_grad_(True) if k.startswith(pre) else v) for k, v in ssd.items()}
                xin = [v.clone().requires_grad_(True) for v in chain]
                with O.bf16_stored_maps():
                    ys = [O.head_tower_layer(leaf, v, branch, i) for v in xin]
                    gs = [bf16_noise(v.shape, 3000 + 10 * i + l, float(v.detach().std())) for l, v in enumerate(ys)]
                    torch.autograd.backward(ys, gs)
                model.zero_grad(set_to_none=True)
                hin = flat(chain).to(K.act_dtype()).cuda().requires_grad_(True)
                hh = Fn.HeadConvGN.apply(hin, m.conv.weight, m.gn.weight, m.gn.bias, sizes, m.gn.eps)
                hh.backward(flat(gs).to(K.act_dtype()).cuda())
                rows.append((f"{branch}_convs.{i}", rel(hh.detach().float().cpu(), flat([v.detach() for v in ys])),
                             rel(hin.grad.float().cpu(), flat([v.grad for v in xin])),
                             _pg_err(_param_grads(model, pre), {k: v.grad for k, v in leaf.items() if k.startswith(pre)}), FWD_TOL))
                chain = [v.detach() for v in ys]
            # the output convolution of the branch (fp32 results) on the tower's last maps
            out_name = "gfl_cls" if branch == "cls" else "gfl_reg"
            keys = [f"bbox_head.{out_name}.weight", f"bbox_head.{out_name}.bias"] + ([f"bbox_head.scales.{l}.scale" for l in range(5)] if branch == "reg" else [])
            leaf = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in ssd.items()}
            xin = [v.clone().requires_grad_(True) for v in chain]
            with O.bf16_stored_maps():
                ys = [O._conv2d(v, leaf[keys[0]], leaf[keys[1]], 1, 1) * (leaf[f"bbox_head.scales.{l}.scale"] if branch == "reg" else 1.0)
                      for l, v in enumerate(xin)]
                gen = torch.Generator().manual_seed(4000)
                gs = [torch.randn(v.shape, generator=gen) * 1e-3 for v in ys]
                torch.autograd.backward(ys, gs)
            model.zero_grad(set_to_none=True)
            hin = flat(chain).to(K.act_dtype()).cuda().requires_grad_(True)
            conv = getattr(model.bbox_head, out_name)
            hh = Fn.HeadConvBias.apply(hin, conv.weight, conv.bias, sizes)
            if branch == "reg":
                hh = Fn.LevelScale.apply(hh, torch.stack([s_.scale for s_ in model.bbox_head.scales]), sizes)
            hh.backward(flat(gs).cuda())
            got = {k: v for k, v in _param_grads(model, "bbox_head.").items() if k in keys}
            rows.append((out_name, rel(hh.detach().float().cpu(), flat([v.detach() for v in ys])),
                         rel(hin.grad.float().cpu(), flat([v.grad for v in xin])), _pg_err(got, {k: leaf[k].grad for k in keys}), FWD_TOL))
        for r in rows:
            print("seed %d, %-16s forward %.2e   input gradient %s   parameter gradients %s"
                  % (seed, r[0], r[1], "-" if r[2] is None else "%.2e" % r[2], "-" if r[3] is None else "%.2e" % r[3]))
        for name, f, dx, dp, ftol in rows:
            assert f <= ftol, (name, f)
            assert dx is None or dx <= BWD_TOL, (name, dx)
            assert dp is None or dp <= BWD_TOL, (name, dp)
        # what the backward bound can see: a unit whose gradients were off by a factor 1.01 would sit at sqrt(err^2 + 1e-4) -- above
        # BWD_TOL for every unit measured here (its own error is >= 1.2e-4 ... and 1e-2 alone already exceeds the bound)
        assert (1e-4) ** 0.5 > BWD_TOL
    finally:
        torch.set_num_threads(threads)


def test_bf16_unitwise_bound_catches_a_one_percent_defect(setup):
    """negative control: the layer3.0 comparison with ONE folded BN scale of the HIP side off by 1 % (a stand-in for a kernel whose
    epilogue scale is wrong) must break the forward bound by a wide margin -- the end-to-end cosine does not notice it (module docstring)"""
    ssd, model = setup
    _, sub, c, _ = _oracle_chain(ssd, 7)
    with torch.no_grad(), O.bf16_stored_maps():
        y = O.resnet_block(sub, c[1], 2, 0)
    blk = model.backbone.layer3[0]
    keep = blk.bn2.weight.detach().clone()
    try:
        with torch.no_grad():
            blk.bn2.weight.mul_(DEFECT)
            bad = rel(from_map(blk(to_map(c[1]))), y)
    finally:
        with torch.no_grad():
            blk.bn2.weight.copy_(keep)
    with torch.no_grad():
        good = rel(from_map(blk(to_map(c[1]))), y)
    print("layer3.0 forward against the oracle: %.2e; with its bn2 scale x %.2f: %.2e" % (good, DEFECT, bad))
    assert good <= FWD_TOL and bad >= 3.0 * FWD_TOL, (good, bad)
