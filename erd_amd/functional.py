"""Differentiable building blocks of the ERD step.  Each ``torch.autograd.Function`` is a fused group of
HIP launches (through ``erd_amd.kernels`` -> C ABI) and implements its own backward with the hand-written
dgrad / wgrad / norm-backward kernels; torch autograd only sequences them (plumbing).

Maps are NHWC fp32 ``[N,H,W,C]``; the head works on level-concatenated ``[N,A,C]`` buffers whose per-level
views are maps.  Weights are ``nn.Parameter``s of logical shape OIHW in channels_last memory, i.e.
``[Cout][kh][kw][Cin]`` as the kernels want them (checkpoint ABI = the reference's OIHW keys/shapes)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import kernels as K

Tensor = torch.Tensor


def ohwi(w: Tensor) -> Tensor:
    """OIHW parameter -> [O,H,W,I] contiguous view (copy only if the parameter is not channels_last)."""
    v = w.detach().permute(0, 2, 3, 1)
    if not v.is_contiguous():
        return v.contiguous()
    v._erd_owner = w      # lets the conv wrappers find per-parameter caches (bf16 twins) without guessing from ._base
    return v


def _to_oihw(dw_ohwi: Tensor) -> Tensor:
    return dw_ohwi.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------------
# fork/join helper: the weight-gradient GEMM of a layer is independent of its input-gradient GEMM, so it is
# enqueued on an auxiliary HIP stream; two MFMA-bound kernels in flight fill each other's ragged dispatch rounds.
# ---------------------------------------------------------------------------------------------------
_AUX = {}
import os as _os
WGRAD_ON_AUX_STREAM = _os.environ.get('ERD_WGRAD_AUX', '0') != '0'
TOWERS_ON_TWO_STREAMS = _os.environ.get('ERD_TOWER_AUX', '1') != '0'


_TOWER = {}

# Stream capture (ERDTrainer(step_graph=True)): the HIP runtime of this ROCm registers the waiting stream as a "parallel
# capture stream" of the event's stream on EVERY hipStreamWaitEvent issued by a stream other than the capture's origin
# (hip::Stream::EndCapture then recurses over those lists).  A join back into a forked stream -- or a stream waiting on
# its own event -- therefore closes a cycle and hipStreamEndCapture recurses until the stack is gone.  Forking from any
# stream is fine, joining is only safe INTO the origin stream.  While a capture is under way CAPTURE_ORIGIN holds the
# origin's handle and the helpers below do not fork from any other stream (the work runs in line there instead).
CAPTURE_ORIGIN = None


def _may_fork(cur) -> bool:
    return CAPTURE_ORIGIN is None or cur.cuda_stream == CAPTURE_ORIGIN


# Trailing weight gradients (backbone backward): the input-gradient chain dz3 -> dz2 -> dz1 -> dx of a bottleneck is the
# critical path of the backward pass, the weight gradients (partial slabs, reduce, d gamma: ~10 launches per block, half of
# them tiny) hang off it.  With TRAIL on they are queued on ONE auxiliary stream that only ever waits for the main stream
# (an event per convolution) and is joined ONCE, at the end of the backward pass (`trail_join`, called by the trainer on
# the stream the optimizer step is queued on) -- the chain never waits for a reduce or a 5-microsecond d gamma launch.
# Only with gradient sinks (ERDTrainer, which owns the join): the results land in the flat gradient buffer, nothing is
# handed back to autograd from the auxiliary stream.
WGRAD_TRAIL = _os.environ.get('ERD_WGRAD_TRAIL', '1') != '0'
RES_LAYER_NODE = _os.environ.get('ERD_RES_LAYER', '1') != '0'    # a whole ResNet stage as one autograd node (ResLayerFn): A/B aid
HEAD_TRAIL = _os.environ.get('ERD_HEAD_TRAIL', '0') != '0'      # the head towers' weight gradients too (A/B aid: see _wgrad_plain)
_TRAIL = {}
_TRAIL_ACTIVE = set()


def trail_stream(device) -> "torch.cuda.Stream":
    key = str(torch.device(device))
    if key not in _TRAIL:
        _TRAIL[key] = torch.cuda.Stream(device=device)
    return _TRAIL[key]


def trail_join(device=None) -> None:
    """the current stream waits for every trailing weight gradient queued so far"""
    for key in list(_TRAIL_ACTIVE):
        if device is None or key == str(torch.device(device)):
            torch.cuda.current_stream(torch.device(key)).wait_stream(_TRAIL[key])
            _TRAIL_ACTIVE.discard(key)


class _Trail:
    """`with _Trail(device, tensors...)`: the body runs on the trailing stream, ordered behind everything queued on the
    current stream so far; the listed tensors (inputs allocated on the current stream) stay alive for it."""

    def __init__(self, device, *tensors):
        self.cur = torch.cuda.current_stream(device)
        self.aux = trail_stream(device)
        self.tensors = [t for t in tensors if t is not None]
        self.key = str(torch.device(device))
        self.inline = not _may_fork(self.cur)

    def __enter__(self):
        if self.inline:
            return self
        self.aux.wait_stream(self.cur)
        for t in self.tensors:
            t.record_stream(self.aux)
        _TRAIL_ACTIVE.add(self.key)
        self.ctx = torch.cuda.stream(self.aux)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if not self.inline:
            self.ctx.__exit__(*a)
        return False


def aux_stream(device) -> "torch.cuda.Stream":
    """the auxiliary stream paired with the current stream (used for the head's second tower)"""
    cur = torch.cuda.current_stream(device)
    if not TOWERS_ON_TWO_STREAMS or not _may_fork(cur):
        return cur
    key = (str(device), cur.cuda_stream)
    if key not in _TOWER:
        _TOWER[key] = torch.cuda.Stream(device=device)
    return _TOWER[key]


class _Fork:
    def __init__(self, device):
        self.cur = torch.cuda.current_stream(device)
        key = (str(device), self.cur.cuda_stream)
        if key not in _AUX:
            _AUX[key] = torch.cuda.Stream(device=device)
        self.aux = _AUX[key] if WGRAD_ON_AUX_STREAM and _may_fork(self.cur) else None

    def __enter__(self):
        if self.aux is not None:
            self.aux.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.aux)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.aux is not None:
            self.ctx.__exit__(*a)
        return False

    def join(self):
        if self.aux is not None:
            self.cur.wait_stream(self.aux)


# ---------------------------------------------------------------------------------------------------------
# gradient sinks: under ERDTrainer every trainable parameter's .grad is a view into ONE flat, per-step zeroed buffer
# (engine.FlatParams marks them `_erd_sink`).  The backward kernels then write the gradient straight into that slot
# and return None to autograd: no temporary, no AccumulateGrad add launch (~160 per step).  The data-parallel bucket
# hooks are told by hand (`_erd_sink_notify`).  Contract: the slot is zero when backward starts and each parameter is
# used by one node per step -- ERDTrainer.zero_grad() every step guarantees both.
# ---------------------------------------------------------------------------------------------------------
def _sink(p: Optional[Tensor]) -> Optional[Tensor]:
    if p is not None and getattr(p, "_erd_sink", False) and p.grad is not None:
        return p.grad
    return None


def _sunk(p: Tensor) -> None:
    n = getattr(p, "_erd_sink_notify", None)
    if n is not None:
        n(p)


def _emit_wgrad(w: Tensor, part: Tensor, S: int, wk: Tensor, scale: Optional[Tensor], rowdot: Optional[Tensor]):
    """reduce the split-K partial slabs into the weight gradient: into the flat slot (returns None) or a new tensor.
    `rowdot` comes from K.zeros_f32 (already zero)."""
    sink = _sink(w)
    if sink is not None:
        K.wgrad_reduce(part, S, wk, scale, ohwi(sink), True, rowdot, rowdot_zeroed=True)
        _sunk(w)
        return None
    dWk = torch.empty_like(wk)
    K.wgrad_reduce(part, S, wk, scale, dWk, False, rowdot, rowdot_zeroed=True)
    return _to_oihw(dWk)


def _bn_fold_cached(gamma, beta, mean, var, eps):
    """folded (scale, shift) of a frozen-statistics BN; cached while gamma/beta are frozen too (teacher, stem,
    layer1).  The cache lives ON the parameter object (dies with it -- a pointer-keyed table would hand a stale
    entry to a new tensor that happens to reuse the address) and is validated against the storage pointers and
    in-place version counters of all four tensors, so loading a checkpoint / re-homing the data invalidates it."""
    if gamma.requires_grad or beta.requires_grad:
        pre = getattr(gamma, "_erd_prefold", None)      # folded for this step by the trainer's batched launch (BnPrefold)?
        if pre is not None and pre[0] == (gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma._version,
                                         beta._version, mean._version, var._version, eps) and pre[1][0]:
            return pre[2], pre[3]
        return K.bn_fold(gamma.detach(), beta.detach(), mean, var, eps)
    ver = (gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(),
           gamma._version, beta._version, mean._version, var._version, eps)
    hit = getattr(gamma, "_erd_fold", None)
    if hit is None or hit[0] != ver:
        hit = (ver, K.bn_fold(gamma.detach(), beta.detach(), mean, var, eps))
        gamma._erd_fold = hit
    return hit[1]


class BnPrefold:
    """All trainable frozen-statistics BNs of a model folded in ONE launch (erd_bn_fold_batch).  ERDTrainer calls `run()`
    right after every optimizer update; until the next update `_bn_fold_cached` hands out views of the result instead of
    launching a 5-microsecond kernel per BN on the forward pass's critical path.  The entry on each gamma is validated
    like the frozen-BN cache (storage pointers + in-place versions); `valid` is cleared when the parameters change
    behind the trainer's back (the trainer re-arms it after its own raw-pointer update)."""

    def __init__(self, model):
        from ._lib import BnFoldItem
        bns = [m for m in model.modules() if hasattr(m, "running_var") and hasattr(m, "weight") and m.weight is not None
               and (m.weight.requires_grad or m.bias.requires_grad)]
        self.bns = bns
        self.valid = [False]
        if not bns:
            return
        dev = bns[0].weight.device
        total = sum(m.weight.numel() for m in bns)
        self.buf = torch.empty(2 * total, dtype=torch.float32, device=dev)
        items = (BnFoldItem * len(bns))()
        off = 0
        self.views = []
        for it, m in zip(items, bns):
            n = m.weight.numel()
            sc, sh = self.buf[off:off + n], self.buf[total + off:total + off + n]
            it.gamma, it.beta, it.mean, it.var = m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr()
            it.scale, it.shift, it.n, it.eps = sc.data_ptr(), sh.data_ptr(), n, float(m.eps)
            sc._erd_stable = True           # (kernels.ParamPrep: a row scale at a fixed address, refreshed every step)
            self.views.append((sc, sh))
            off += n
        import ctypes as C
        raw = bytes(memoryview(items))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.max_n = max(m.weight.numel() for m in bns)
        self._ptrs = None

    def run(self) -> None:
        if not self.bns:
            return
        self._run(self.bns, self.views, self.table, self.max_n, None)

    def set_groups(self, group_of) -> None:
        """`group_of(gamma parameter) -> int`: run(group) folds only that group's BNs (ERDTrainer: per gradient bucket)"""
        from ._lib import BnFoldItem
        import ctypes as C
        size = C.sizeof(BnFoldItem)
        self.groups, self.gtables, self._gptrs = {}, {}, {}
        if not self.bns:
            return
        for j, m in enumerate(self.bns):
            g = group_of(m.weight)
            self.groups.setdefault(group_of(m.bias) if g is None else g, []).append(j)
        host = self.table.cpu()
        for g, js in self.groups.items():
            sub = torch.cat([host[j * size:(j + 1) * size] for j in js]).to(self.table.device)
            self.gtables[g] = (sub, [self.bns[j] for j in js], [self.views[j] for j in js], max(self.bns[j].weight.numel() for j in js))

    def run_group(self, g: int) -> None:
        ent = self.gtables.get(g)
        if ent is None:
            return
        table, bns, views, max_n = ent
        self._run(bns, views, table, max_n, g)

    def _run(self, bns, views, table, max_n, g) -> None:
        ptrs = tuple((m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr()) for m in bns)
        seen = self._ptrs if g is None else self._gptrs.get(g)
        if seen is not None and ptrs != seen:
            self.valid[0] = False          # re-homed parameters: the device table is stale -> fall back to per-call folds
            return
        if g is None:
            self._ptrs = ptrs
        else:
            self._gptrs[g] = ptrs
        K.call("erd_bn_fold_batch", K._p(table), len(bns), max_n, K._stream())
        for m, (sc, sh) in zip(bns, views):
            g_, b = m.weight, m.bias
            g_._erd_prefold = ((g_.data_ptr(), b.data_ptr(), m.running_mean.data_ptr(), m.running_var.data_ptr(), g_._version, b._version,
                                m.running_mean._version, m.running_var._version, m.eps), self.valid, sc, sh)
        self.valid[0] = True


def _wgrad_plain(w: Tensor, xs, dzs, k: int, stride: int, pad: int, keep=(), trail: bool = True):
    """weight gradient of a convolution without BN coupling: on the trailing stream straight into the parameter's flat
    gradient slot (-> None) when it has one, else in place on the current stream (-> the gradient tensor).  `keep`: the
    allocations behind the views in xs / dzs (kept alive for the trailing stream).  `trail=False`: the head towers --
    their 0.9 ms weight gradients already alternate with the input gradients on the two tower streams; queued behind each
    other on the ONE trailing stream they become a 7 ms serial tail that the join at the end of backward waits for
    (measured: 80.0 img/s with them trailing, 80.3 without)."""
    wk = ohwi(w)
    if trail and WGRAD_TRAIL and _sink(w) is not None:
        with _Trail(xs[0].device, *(keep or (list(xs) + list(dzs)))):
            part, S = K.conv_wgrad_partials(xs, dzs, k, stride, pad)
            return _emit_wgrad(w, part, S, wk, None, None)
    part, S = K.conv_wgrad_partials(xs, dzs, k, stride, pad)
    return _emit_wgrad(w, part, S, wk, None, None)


class ConvBNAct(Function):
    """y = [relu]( conv(x, w) * scale + shift [+ res] ) with scale/shift folded from a frozen-statistics BN
    (resnet.py:263-302 Bottleneck; norm_eval=True resnet.py:648-657).  gamma/beta still receive gradients
    (d gamma through the <W, G> identity, see erd_hip.h erd_wgrad_reduce)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, mean, var, res, k: int, stride: int, pad: int, relu: bool, eps: float):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        wk = ohwi(w)
        scale, shift = _bn_fold_cached(gamma, beta, mean, var, eps)
        N, H, W_, _ = x.shape
        OH, OW = K.conv_out_size(H, k, stride, pad), K.conv_out_size(W_, k, stride, pad)
        out = torch.empty((N, OH, OW, wk.shape[0]), dtype=x.dtype, device=x.device)      # maps keep their storage type
        K.conv_forward([x], wk, [out], k, stride, pad, scale=scale, shift=shift,
                       res=None if res is None else [res], relu=relu)
        ctx.cfg = (k, stride, pad, relu, eps, res is not None)
        ctx.save_for_backward(x, w, mean, var, scale, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, var, scale, out = ctx.saved_tensors
        k, stride, pad, relu, eps, has_res = ctx.cfg
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[0:4]
        dy = dy.contiguous()
        dz, dbeta = K.relu_bwd_colsum(out if relu else None, dy, relu, want_colsum=need_g or need_b)
        wk = ohwi(w)
        dW = dgamma = None
        fork = _Fork(x.device)
        if need_w or need_g:
            rowdot = K.zeros_f32(scale.numel(), scale.device) if need_g else None
            dgamma = torch.empty_like(scale) if need_g else None
            with fork:
                part, S = K.conv_wgrad_partials([x], [dz], k, stride, pad)
                dW = _emit_wgrad(w, part, S, wk, scale, rowdot)
                if need_g:
                    K.bn_dgamma(rowdot, dbeta, mean, var, eps, out=dgamma)
        dx = None
        if need_x:
            wt = K.weight_transpose(wk, scale)
            # a strided 1x1 (projection shortcut) reaches only the even pixels: the rest of dx is zero
            dx = torch.zeros_like(x) if k < stride else torch.empty_like(x)
            K.conv_dgrad([dz], wt, [dx], k, stride, pad)
        fork.join()
        dres = dz if (has_res and ctx.needs_input_grad[6]) else None
        return dx, dW, dgamma, (dbeta if need_b else None), None, None, dres, None, None, None, None, None


class BottleneckFn(Function):
    """A whole ResNet bottleneck (resnet.py:263-302) as ONE autograd node with a hand-scheduled backward:
        o1 = relu(bn1(conv1 x)); o2 = relu(bn2(conv2 o1)); y = relu(bn3(conv3 o2) + shortcut(x))
    Backward: only y's ReLU needs its own pass (its gradient has several producers); the masks of o2 / o1 and the
    d-beta column sums are applied in the EPILOGUE of the input-gradient GEMM that produces their gradient, and the
    identity shortcut's gradient is the residual operand of conv1's input-gradient GEMM -- no stand-alone ReLU
    backward, no gradient-accumulation kernels inside the block."""

    @staticmethod
    def forward(ctx, x, stride: int, eps: float, *params):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        # params: (w,g,b,mean,var) x {conv1,conv2,conv3[,downsample]}
        has_down = len(params) == 20
        P = [params[5 * i:5 * i + 5] for i in range(4 if has_down else 3)]
        dev = x.device

        def cba(inp, prm, k, s, pad, res, relu):
            w, g, b, m, v = prm
            wk = ohwi(w)
            scale, shift = _bn_fold_cached(g, b, m, v, eps)
            N, H, W_, _ = inp.shape
            out = torch.empty((N, K.conv_out_size(H, k, s, pad), K.conv_out_size(W_, k, s, pad), wk.shape[0]),
                              dtype=inp.dtype, device=dev)
            K.conv_forward([inp], wk, [out], k, s, pad, scale=scale, shift=shift,
                           res=None if res is None else [res], relu=relu)
            return out, scale

        o1, s1 = cba(x, P[0], 1, 1, 0, None, True)
        o2, s2 = cba(o1, P[1], 3, stride, 1, None, True)
        sd = None
        idn = x
        if has_down:
            idn, sd = cba(x, P[3], 1, stride, 0, None, False)
        y, s3 = cba(o2, P[2], 1, 1, 0, idn, True)
        ctx.cfg = (stride, eps, has_down)
        ctx.save_for_backward(x, o1, o2, y, s1, s2, s3, *( [sd] if has_down else []), *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, eps, has_down = ctx.cfg
        saved = ctx.saved_tensors
        x, o1, o2, y, s1, s2, s3 = saved[:7]
        off = 7
        sd = None
        if has_down:
            sd = saved[7]
            off = 8
        params = saved[off:]
        P = [params[5 * i:5 * i + 5] for i in range(4 if has_down else 3)]
        need = ctx.needs_input_grad
        b3 = P[2][2]
        s_ = _sink(b3) if need[3 + 5 * 2 + 2] else None
        dz3, db3 = K.relu_bwd_colsum(y, dy.contiguous(), True, colsum_into=s_ if s_ is not None else K.zeros_f32(y.shape[3], x.device))
        dx, grads = _bottleneck_backward(x, o1, o2, s1, s2, s3, sd, P, list(need[3:]), need[0], stride, eps, dz3, db3)
        return (dx, None, None, *grads)


def _bottleneck_backward(x, o1, o2, s1, s2, s3, sd, P, need_p, need_x, stride, eps, dz3, db3, out_mask=None):
    """Backward of one bottleneck given dz3 = dy * (y > 0) and its column sums db3 (a [C] vector that may BE d beta's flat slot,
    or a replicated [copies, C] accumulator of a gradient convolution's epilogue).  need_p: needs-gradient flags of the block's
    parameters, 5 per convolution (w, gamma, beta, mean, var).  Returns (dx, grads).  With `out_mask` (the block's own input x,
    which is the previous block's post-ReLU output) the last input-gradient launch ALSO applies that ReLU's mask and column-sums
    the result into a fresh replicated accumulator: it then returns ((dz3 of the previous block, its db3), grads) -- the
    stand-alone ReLU-backward pass between two blocks of a stage disappears (identity-shortcut blocks only)."""
    has_down = sd is not None
    dev = x.device
    grads = [None] * (5 * len(P))

    def wgrad(idx, xin, dz, k, s, pad, scale, dbeta):
        """dW (scaled by the folded BN), d gamma, d beta of conv `idx`"""
        w, g, b, m, v = P[idx]
        base = 5 * idx
        if not (need_p[base] or need_p[base + 1] or need_p[base + 2]):
            return
        wk = ohwi(w)
        trail = None
        if WGRAD_TRAIL and all((not need_p[base + q]) or _sink(t) is not None for q, t in enumerate((w, g, b))):
            trail = _Trail(dev, xin, dz, scale, dbeta)
            trail.__enter__()
        try:
            _wgrad_body(idx, xin, dz, k, s, pad, scale, dbeta, w, g, b, m, v, base, wk)
        finally:
            if trail is not None:
                trail.__exit__(None, None, None)

    def _wgrad_body(idx, xin, dz, k, s, pad, scale, dbeta, w, g, b, m, v, base, wk):
        part, S = K.conv_wgrad_partials([xin], [dz], k, s, pad)
        rowdot = K.zeros_f32(scale.numel(), scale.device) if need_p[base + 1] else None
        grads[5 * idx] = _emit_wgrad(w, part, S, wk, scale, rowdot)
        if need_p[base + 1]:
            # a replicated [copies, C] accumulator is folded here; the fold lands in d beta's flat slot (zero at
            # this point) or in a new tensor
            rep = dbeta.dim() == 2
            bs_fold = (_sink(b) if need_p[base + 2] else None) if rep else None
            fold = None if not rep else (bs_fold if bs_fold is not None else torch.empty_like(dbeta[0]))
            gs = _sink(g)
            if gs is not None:                      # the slot is zero: plain store == accumulation
                K.bn_dgamma(rowdot, dbeta, m, v, eps, out=gs, dbeta_out=fold)
                _sunk(g)
            else:
                grads[5 * idx + 1] = K.bn_dgamma(rowdot, dbeta, m, v, eps, dbeta_out=fold)
            if rep:
                dbeta = fold
        elif dbeta.dim() == 2:
            dbeta = dbeta.sum(0)
        if need_p[base + 2]:
            bs = _sink(b)
            if bs is None:
                grads[5 * idx + 2] = dbeta
            else:
                if dbeta.data_ptr() != bs.data_ptr():   # (the projection shortcut shares conv3's column sums)
                    bs.add_(dbeta)
                _sunk(b)

    wgrad(2, o2, dz3, 1, 1, 0, s3, db3)
    dz2 = torch.empty_like(o2)
    # the epilogues of the two input-gradient convolutions add their column sums into 8-128 replicated rows (hundreds of
    # tiles adding to ONE row would serialise on the memory-side atomic unit: K.colsum_copies); wgrad()'s bn_dgamma folds them
    rep_slot = lambda n: K.zeros_f32(K.colsum_copies(n) * n, dev).view(K.colsum_copies(n), n)
    db2 = rep_slot(o2.shape[3])
    K.conv_dgrad([dz3], K.weight_transpose(ohwi(P[2][0]), s3), [dz2], 1, 1, 0, relu_mask=[o2], colsum=db2)
    wgrad(1, o1, dz2, 3, stride, 1, s2, db2)
    dz1 = torch.empty_like(o1)
    db1 = rep_slot(o1.shape[3])
    K.conv_dgrad([dz2], K.weight_transpose(ohwi(P[1][0]), s2), [dz1], 3, stride, 1, relu_mask=[o1], colsum=db1)
    wgrad(0, x, dz1, 1, 1, 0, s1, db1)
    dx = None
    if need_x:
        dx = torch.empty_like(x)
        wt1 = K.weight_transpose(ohwi(P[0][0]), s1)
        if has_down:
            assert out_mask is None, "the fused ReLU mask needs an identity shortcut (one launch writes every pixel of dx)"
            K.conv_dgrad([dz1], wt1, [dx], 1, 1, 0)
            K.conv_dgrad([dz3], K.weight_transpose(ohwi(P[3][0]), sd), [dx], 1, stride, 0, accumulate=True)
        elif out_mask is not None:      # + identity gradient, the previous block's ReLU mask and its d-beta column sums, in the epilogue
            db_prev = rep_slot(x.shape[3])
            K.conv_dgrad([dz1], wt1, [dx], 1, 1, 0, res=[dz3], relu_mask=[out_mask], colsum=db_prev)
            dx = (dx, db_prev)
        else:
            K.conv_dgrad([dz1], wt1, [dx], 1, 1, 0, res=[dz3])      # + identity gradient, in the epilogue
    if has_down:
        wgrad(3, x, dz3, 1, stride, 0, sd, db3)
    return dx, grads


class ResLayerFn(Function):
    """A whole ResNet stage (res_layer.py:57-63: one projection block + identity blocks) as ONE autograd node.  Forward: the
    launches of the blocks' BottleneckFn, one after the other.  Backward: the blocks' hand-scheduled backward passes chained
    directly -- the gradient a block hands to its predecessor is produced by ONE input-gradient launch that already carries
    the predecessor's output-ReLU mask and the column sums for its d beta, so only the LAST block of the stage (whose output
    also leaves the stage) runs a stand-alone ReLU backward: 3 instead of 13 such passes per step for ResNet-50, and 3 autograd
    nodes instead of 13."""

    @staticmethod
    def forward(ctx, x, strides, eps: float, counts, *params):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        dev = x.device
        blocks, off = [], 0
        for n in counts:                             # parameters per block: 15 (identity shortcut) or 20 (projection)
            blocks.append(params[off:off + n])
            off += n
        saved, meta = [], []

        def cba(inp, prm, k, s, pad, res, relu):
            w, g, b, m, v = prm
            wk = ohwi(w)
            scale, shift = _bn_fold_cached(g, b, m, v, eps)
            N, H, W_, _ = inp.shape
            out = torch.empty((N, K.conv_out_size(H, k, s, pad), K.conv_out_size(W_, k, s, pad), wk.shape[0]),
                              dtype=inp.dtype, device=dev)
            K.conv_forward([inp], wk, [out], k, s, pad, scale=scale, shift=shift,
                           res=None if res is None else [res], relu=relu)
            return out, scale

        h = x
        for bp, stride in zip(blocks, strides):
            has_down = len(bp) == 20
            P = [bp[5 * i:5 * i + 5] for i in range(4 if has_down else 3)]
            o1, s1 = cba(h, P[0], 1, 1, 0, None, True)
            o2, s2 = cba(o1, P[1], 3, stride, 1, None, True)
            sd, idn = None, h
            if has_down:
                idn, sd = cba(h, P[3], 1, stride, 0, None, False)
            y, s3 = cba(o2, P[2], 1, 1, 0, idn, True)
            meta.append((stride, has_down, len(saved)))
            saved += [h, o1, o2, y, s1, s2, s3] + ([sd] if has_down else [])
            h = y
        ctx.meta, ctx.eps, ctx.counts, ctx.nsaved = meta, eps, counts, len(saved)
        ctx.save_for_backward(*saved, *params)
        return h

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        acts, params = saved[:ctx.nsaved], saved[ctx.nsaved:]
        need = ctx.needs_input_grad
        blocks, need_b, off = [], [], 0
        for n in ctx.counts:
            blocks.append(params[off:off + n])
            need_b.append(list(need[4 + off:4 + off + n]))
            off += n
        grads_all = [None] * len(params)
        nb = len(blocks)
        dz3 = db3 = None
        dx = None
        for i in range(nb - 1, -1, -1):
            stride, has_down, a0 = ctx.meta[i]
            x, o1, o2, y, s1, s2, s3 = acts[a0:a0 + 7]
            sd = acts[a0 + 7] if has_down else None
            bp = blocks[i]
            P = [bp[5 * j:5 * j + 5] for j in range(4 if has_down else 3)]
            if i == nb - 1:       # the stage's output gradient arrives from outside: the one stand-alone ReLU backward
                b3 = P[2][2]
                s_ = _sink(b3) if need_b[i][5 * 2 + 2] else None
                dz3, db3 = K.relu_bwd_colsum(y, dy.contiguous(), True,
                                             colsum_into=s_ if s_ is not None else K.zeros_f32(y.shape[3], x.device))
            # block i hands block i-1 its dz3 directly when block i has an identity shortcut (x IS block i-1's ReLU output)
            fuse = i > 0 and not has_down
            need_x = need[0] if i == 0 else True
            out, g = _bottleneck_backward(x, o1, o2, s1, s2, s3, sd, P, need_b[i], need_x, stride, ctx.eps, dz3, db3,
                                          out_mask=x if fuse else None)
            base = sum(ctx.counts[:i])
            grads_all[base:base + len(g)] = g
            if i == 0:
                dx = out
            elif fuse:
                dz3, db3 = out
            else:                  # (a projection block in the middle of a stage: not a ResNet layout, kept general)
                prev_y = acts[ctx.meta[i - 1][2] + 3]
                pb3 = blocks[i - 1][5 * 2 + 2]
                s_ = _sink(pb3) if need_b[i - 1][5 * 2 + 2] else None
                dz3, db3 = K.relu_bwd_colsum(prev_y, out, True,
                                             colsum_into=s_ if s_ is not None else K.zeros_f32(prev_y.shape[3], x.device))
        return (dx, None, None, None, *grads_all)


def _bias_grad(b: Tensor, dy: Tensor, rows: Optional[Tensor] = None):
    """column sums of dy as the gradient of bias `b`: straight into the parameter's flat gradient slot (-> None, the slot is
    zero and each parameter is used by one node per step) or as a new tensor for autograd to accumulate"""
    sink = _sink(b)
    if sink is not None:
        K.colsum(dy if rows is None else rows, out=sink)
        _sunk(b)
        return None
    return K.colsum(dy if rows is None else rows)


class ConvBias(Function):
    """y = conv(x, w) + b on one map (FPN laterals, fpn.py:177-179)."""

    @staticmethod
    def forward(ctx, x, w, b, k: int, stride: int, pad: int):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        wk = ohwi(w)
        N, H, W_, _ = x.shape
        OH, OW = K.conv_out_size(H, k, stride, pad), K.conv_out_size(W_, k, stride, pad)
        out = torch.empty((N, OH, OW, wk.shape[0]), dtype=x.dtype, device=x.device)
        K.conv_forward([x], wk, [out], k, stride, pad, shift=b.detach())
        ctx.cfg = (k, stride, pad)
        ctx.bias_param = b
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, stride, pad = ctx.cfg
        dy = dy.contiguous()
        wk = ohwi(w)
        dW = db = dx = None
        if ctx.needs_input_grad[1]:
            dW = _wgrad_plain(w, [x], [dy], k, stride, pad)
        if ctx.needs_input_grad[2]:
            db = _bias_grad(ctx.bias_param, dy)
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x) if k < stride else torch.empty_like(x)
            K.conv_dgrad([dy], K.weight_transpose(wk), [dx], k, stride, pad)
        return dx, dW, db, None, None, None


class UpsampleAdd(Function):
    """fine += nearest_upsample(coarse) in place (fpn.py:181-191)."""

    @staticmethod
    def forward(ctx, fine, coarse):
        ctx.mark_dirty(fine)
        K.upsample_add_(fine, coarse)
        ctx.cshape = coarse.shape
        return fine

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dcoarse = None
        if ctx.needs_input_grad[1]:
            dcoarse = torch.zeros(ctx.cshape, dtype=dout.dtype, device=dout.device)
            K.upsample_add_bwd_(dout, dcoarse)
        return dout, dcoarse


class FPNOutputs(Function):
    """The five FPN output convs (3x3 on the three merged laterals, then P6 = 3x3/2 on P5's OUTPUT and
    P7 = 3x3/2 on P6, no activation: fpn.py:195-220) writing straight into one level-concatenated
    [N,A,256] buffer -- the layout the head, ERS and the losses consume (K10 "free" permute+cat)."""

    @staticmethod
    def forward(ctx, l3, l4, l5, w0, w1, w2, w3, w4, b0, b1, b2, b3, b4):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        lats = [l3, l4, l5]
        ws = [w0, w1, w2, w3, w4]
        bs = [b0, b1, b2, b3, b4]
        N = l3.shape[0]
        sizes = [(l.shape[1], l.shape[2]) for l in lats]
        h, w_ = sizes[-1]
        for _ in range(2):
            h, w_ = K.conv_out_size(h, 3, 2, 1), K.conv_out_size(w_, 3, 2, 1)
            sizes.append((h, w_))
        A = sum(a * b for a, b in sizes)
        Cc = w0.shape[0]
        cat = torch.empty((N, A, Cc), dtype=l3.dtype, device=l3.device)
        views = K.level_views(cat, sizes)
        for i in range(3):
            K.conv_forward([lats[i]], ohwi(ws[i]), [views[i]], 3, 1, 1, shift=bs[i].detach())
        K.conv_forward([views[2]], ohwi(ws[3]), [views[3]], 3, 2, 1, shift=bs[3].detach())
        K.conv_forward([views[3]], ohwi(ws[4]), [views[4]], 3, 2, 1, shift=bs[4].detach())
        ctx.sizes = sizes
        ctx.bias_params = bs
        ctx.save_for_backward(l3, l4, l5, cat, *ws)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        l3, l4, l5, cat, *ws = ctx.saved_tensors
        sizes = ctx.sizes
        lats = [l3, l4, l5]
        # The P5 / P6 slices accumulate the extra-level input gradients.  The incoming tensor belongs to the caller (a user-supplied
        # grad_outputs, a retained gradient, a hook's capture: an autograd.Function must not modify its gradient inputs -- ADVICE
        # r3), so those two slices are accumulated into PRIVATE copies (5.4 MB at the benchmark's size; round 3 skipped the
        # copy altogether, round 2 cloned all 92 MB)
        dcat = dcat.contiguous()
        dv = list(K.level_views(dcat, sizes))
        dv[3] = dv[3].clone()
        dv[2] = dv[2].clone()
        pv = K.level_views(cat, sizes)
        grads_w: List[Optional[Tensor]] = [None] * 5
        grads_b: List[Optional[Tensor]] = [None] * 5

        def wgrad(i, xin, dz, stride):
            wk = ohwi(ws[i])
            if ctx.needs_input_grad[3 + i]:
                grads_w[i] = _wgrad_plain(ws[i], [xin], [dz], 3, stride, 1, keep=(xin, dcat, dz))
            if ctx.needs_input_grad[8 + i]:
                sink = _sink(ctx.bias_params[i])
                if sink is not None:          # column sums straight into the bias' flat gradient slot (zero at this point)
                    K.relu_bwd_colsum(None, dz, False, colsum_into=sink)
                    _sunk(ctx.bias_params[i])
                else:
                    grads_b[i] = K.relu_bwd_colsum(None, dz, False)[1]
            return wk

        # P7 <- P6 output ; P6 <- P5 output (input grads accumulate into the producer's slice)
        wk = wgrad(4, pv[3], dv[4], 2)
        K.conv_dgrad([dv[4]], K.weight_transpose(wk), [dv[3]], 3, 2, 1, accumulate=True)
        wk = wgrad(3, pv[2], dv[3], 2)
        K.conv_dgrad([dv[3]], K.weight_transpose(wk), [dv[2]], 3, 2, 1, accumulate=True)
        dl: List[Optional[Tensor]] = [None, None, None]
        for i in range(3):
            wk = wgrad(i, lats[i], dv[i], 1)
            if ctx.needs_input_grad[i]:
                dl[i] = torch.empty_like(lats[i])
                K.conv_dgrad([dv[i]], K.weight_transpose(wk), [dl[i]], 3, 1, 1)
        return (dl[0], dl[1], dl[2], *grads_w, *grads_b)


class HeadConvGN(Function):
    """One tower layer of the GFL head on all five levels at once (weights shared across levels,
    gfl_head.py:156-177,219-223): conv3x3 (no bias) -> GroupNorm(32) -> ReLU."""

    @staticmethod
    def forward(ctx, x_cat, w, gamma, beta, sizes, eps: float):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        c, y, mr = K.conv3x3_gn_relu_forward(x_cat, ohwi(w), gamma.detach(), beta.detach(), sizes, 32, eps)
        ctx.sizes = sizes
        ctx.save_for_backward(x_cat, w, gamma, beta, c, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cat, w, gamma, beta, c, mr = ctx.saved_tensors
        sizes = ctx.sizes
        gs, bs_ = _sink(gamma), _sink(beta)
        direct = gs is not None and bs_ is not None      # accumulate straight into the flat gradient slots (zero at this point)
        dc, dgamma, dbeta = K.gn_relu_backward(c, dy.contiguous(), gamma.detach(), beta.detach(), mr, sizes, 32,
                                               dgamma=gs if direct else None, dbeta=bs_ if direct else None)
        if direct:
            _sunk(gamma)
            _sunk(beta)
            dgamma = dbeta = None
        wk = ohwi(w)
        dW = dx = None
        xv, dv = K.level_views(x_cat, sizes), K.level_views(dc, sizes)
        if ctx.needs_input_grad[1]:
            dW = _wgrad_plain(w, xv, dv, 3, 1, 1, keep=(x_cat, dc), trail=HEAD_TRAIL)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x_cat)
            K.conv_dgrad(dv, K.weight_transpose(wk), K.level_views(dx, sizes), 3, 1, 1)
        return dx, dW, dgamma, dbeta, None, None


class HeadConvBias(Function):
    """gfl_cls / gfl_reg: conv3x3 + bias on all levels into a [N,A,Cout] buffer (gfl_head.py:224-229)."""

    @staticmethod
    def forward(ctx, x_cat, w, b, sizes):
        K.RECORDED = any(ctx.needs_input_grad)      # see kernels.WINO_TRAIN_FWD
        wk = ohwi(w)
        # the head outputs feed the losses, ERS and NMS: always fp32 (gfl_head.py:224-229; autocast keeps the losses fp32)
        out = torch.empty((x_cat.shape[0], x_cat.shape[1], wk.shape[0]), dtype=torch.float32, device=x_cat.device)
        K.conv_forward(K.level_views(x_cat, sizes), wk, K.level_views(out, sizes), 3, 1, 1, shift=b.detach())
        ctx.sizes = sizes
        ctx.bias_param = b
        ctx.save_for_backward(x_cat, w)
        return out

    @staticmethod
    def backward(ctx, dy):
        x_cat, w = ctx.saved_tensors
        sizes = ctx.sizes
        dy = dy.contiguous()
        wk = ohwi(w)
        dW = db = dx = None
        xv, dv = K.level_views(x_cat, sizes), K.level_views(dy, sizes)
        if ctx.needs_input_grad[1]:
            dW = _wgrad_plain(w, xv, dv, 3, 1, 1, keep=(x_cat, dy), trail=HEAD_TRAIL)
        if ctx.needs_input_grad[2]:
            db = _bias_grad(ctx.bias_param, dy)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x_cat)
            K.conv_dgrad(dv, K.weight_transpose(wk), K.level_views(dx, sizes), 3, 1, 1)
        return dx, dW, db, None


class LevelScale(Function):
    """bbox_pred_l = Scale_l(gfl_reg(.)) -- one learnable scalar per level (gfl_head.py:184,229)."""

    @staticmethod
    def forward(ctx, x_cat, alphas, sizes):
        y = K.level_scale(x_cat, alphas.detach().contiguous(), sizes)
        ctx.sizes = sizes
        ctx.save_for_backward(x_cat, alphas)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cat, alphas = ctx.saved_tensors
        dx, dal = K.level_scale_bwd(x_cat, dy.contiguous(), alphas.detach().contiguous(), ctx.sizes)
        return dx, dal, None


class ERDLossFn(Function):
    """All loss entries of the step as ONE vector
        [loss_cls(L) | loss_bbox(L) | loss_dfl(L) | loss_dist_cls(N) | loss_dist_bbox(N)]
    (gfl_head_increment_erd.py:334-454).  `t` carries the no-grad side: targets, teacher outputs, ERS lists,
    NMS keep mask (None for the plain GFL head loss)."""

    @staticmethod
    def forward(ctx, s_cls, s_bbox, t):
        N, A, c_all = s_cls.shape
        L = len(t.sizes)
        score, wt, sums = K.gfl_losses_fwd(s_cls, s_bbox, t.anchors, t.labels, t.label_weights, t.bbox_targets,
                                           t.sizes, t.strides, t.c_old, c_all)
        avg = K.loss_avg(t.num_pos, sums)
        if t.world_size > 1:       # reduce_mean x2 (dist_utils.py:59-65) fused into one 2-float all-reduce
            from .dist_utils import reduce_mean
            avg = reduce_mean(avg)
        l2s = kds = None
        nd = 0
        if t.distill:
            nd = N
            l2s = K.l2_distill(s_cls, t.t_cls, t.ers["idx_cls"], t.ers["counts"], t.c_old)
            kds = K.kd_kl(s_bbox, t.t_bbox, s_cls, t.keep, t.c_old, t.T)
        losses, _ = K.loss_finalize(sums, avg, l2s, kds, t.ers["counts"] if t.distill else None, L, nd, t.c_old,
                                    t.dist_loss_weight, t.lw_cls, t.lw_bbox, t.lw_dfl, t.lw_ld, None, True, False)
        ctx.t = t
        ctx.nd = nd
        ctx.save_for_backward(s_cls, s_bbox, score, wt, sums, avg, l2s if l2s is not None else sums,
                              kds if kds is not None else sums)
        return losses

    @staticmethod
    def backward(ctx, dlosses):
        s_cls, s_bbox, score, wt, sums, avg, l2s, kds = ctx.saved_tensors
        t, nd = ctx.t, ctx.nd
        N, A, c_all = s_cls.shape
        L = len(t.sizes)
        _, coef = K.loss_finalize(sums, avg, l2s if nd else None, kds if nd else None,
                                  t.ers["counts"] if nd else None, L, nd, t.c_old, t.dist_loss_weight, t.lw_cls,
                                  t.lw_bbox, t.lw_dfl, t.lw_ld, dlosses.contiguous(), False, True)
        dcls, dbbox = K.gfl_losses_bwd(s_cls, s_bbox, t.anchors, t.labels, t.label_weights, t.bbox_targets, t.sizes,
                                       t.strides, t.c_old, c_all, score, wt, coef)
        if nd:
            K.l2_distill_bwd_(s_cls, t.t_cls, t.ers["idx_cls"], t.ers["counts"], coef[4 * L:], t.c_old, dcls)
            K.kd_kl_bwd_(s_bbox, t.t_bbox, s_cls, t.keep, coef[4 * L + N:], t.c_old, t.T, dbbox)
        return dcls, dbbox, None
