"""Thin Python wrappers over the C ABI (``include/erd_hip.h``): they only translate torch
tensors (device memory + current HIP stream = plumbing) into raw pointers / sizes and pick
launch geometry.  No arithmetic happens here and there is no non-HIP fallback."""
from __future__ import annotations

import ctypes as C
import os as _os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, Levels, WgradDesc, call

Tensor = torch.Tensor


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _cur_stream() -> int:
    """hipStream_t of the current stream of the current device as an int (the raw getter skips building a
    torch.cuda.Stream object: ~3 us per call, three calls per convolution launch)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream() -> C.c_void_p:
    return C.c_void_p(_cur_stream())


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_gpu(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.ErdHipError("erd_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")


# ---------------------------------------------------------------------------------------------
# optional per-launch timing of the GEMM-shaped kernels with HIP events recorded on the launch stream
# (bench.py roofline leg).  Off by default: zero overhead in normal runs.
# ---------------------------------------------------------------------------------------------
_TIMING = None


def timing_begin() -> None:
    global _TIMING
    _TIMING = {}


def timing_end():
    """-> {kernel: dict(kernel, ms, flop, launches)} summed over every launch since timing_begin()."""
    global _TIMING
    rec, _TIMING = _TIMING, None
    torch.cuda.synchronize()
    out = {}
    for name, evs in (rec or {}).items():
        ms = sum(s.elapsed_time(e) for s, e, _, _ in evs)
        out[name] = dict(kernel=name, ms=ms, flop=float(sum(f for _, _, f, _ in evs)), launches=len(evs),
                         min_bytes=float(sum(b for _, _, _, b in evs)))
    return out


TIMING_DETAIL = False      # tools/step_breakdown.py: one record per layer shape instead of per kernel class


def _timed_call(kname: str, flop: float, cname: str, *args, nbytes: float = 0.0, tag: str = "") -> None:
    """nbytes: the launch's algorithmic HBM bytes (every operand read once, the result written once)"""
    if _TIMING is None:
        call(cname, *args)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    call(cname, *args)
    e.record()
    if TIMING_DETAIL and tag:
        kname = f"{kname} {tag}"
    _TIMING.setdefault(kname, []).append((s, e, flop, nbytes))


# ---------------------------------------------------------------------------------------------
# workspace cache: one growing buffer per (name, device, stream) -- stream-ordered reuse is safe
# ---------------------------------------------------------------------------------------------
_WS = {}
# A captured hipGraph (engine.TeacherGraphs, ERDTrainer(step_graph=True)) bakes the ADDRESSES of the workspaces its
# kernels used into its nodes.  A later, larger request (another input shape, eager or captured) replaces the buffer; if
# the old allocation were freed, replaying the earlier graph would write into memory the caching allocator has handed to
# somebody else.  From the first capture on (`pin_workspaces()`), outgrown buffers are therefore retired, not freed.
_WS_PINNED = False
_WS_RETIRED: List[Tensor] = []


def pin_workspaces() -> None:
    """called by whoever captures a graph, before the capture: outgrown workspaces stay allocated from now on"""
    global _WS_PINNED
    _WS_PINNED = True


def _ws_replace(key, buf: Tensor) -> None:
    old = _WS.get(key)
    if old is not None and _WS_PINNED:
        _WS_RETIRED.append(old)
    _WS[key] = buf


def workspace(name: str, nbytes: int, device) -> Tensor:
    key = (name, str(device), _cur_stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _ws_replace(key, buf)
    return buf


# ---------------------------------------------------------------------------------------------
# zero arena: the step needs a few hundred small zero-initialised accumulators (column sums, row dots, ...).
# One memset per step over a bump-allocated arena replaces a fill launch per accumulator.
# ---------------------------------------------------------------------------------------------
_ARENA = {"buf": None, "off": 0, "key": None, "ok": ()}


def zero_arena_begin(device, nbytes: int = int(_os.environ.get("ERD_ZERO_ARENA_MB", "16")) << 20, forked_streams=()) -> None:
    """called once per step (on the stream the step runs on) by the trainer.  `forked_streams`: raw handles of streams
    that the step forks to AFTER this call and joins BEFORE zero_arena_end() (the trailing weight-gradient stream, the
    second tower stream): slices may be handed out there too -- they are ordered behind this memset by the fork and the
    next step's memset is ordered behind their last use by the join.  (Round 2 handed those streams torch.zeros: 50 fill
    launches per step for the trailing stream's row-dot / d-gamma scratch.)"""
    if _ARENA["buf"] is None or _ARENA["buf"].device != torch.device(device):
        _ARENA["buf"] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _ARENA["buf"].zero_()
    _ARENA["off"] = 0
    _ARENA["key"] = _cur_stream()
    _ARENA["ok"] = (_ARENA["key"],) + tuple(forked_streams)


def zeros_f32(n: int, device) -> Tensor:
    """n zeroed floats: a slice of the step's arena when one is active on this stream, else torch.zeros"""
    a = _ARENA
    nb = (n * 4 + 255) // 256 * 256
    if a["buf"] is None or a["key"] is None or _cur_stream() not in a["ok"] or a["off"] + nb > a["buf"].numel() \
            or a["buf"].device != torch.device(device):
        return torch.zeros(n, dtype=torch.float32, device=device)
    out = a["buf"][a["off"]:a["off"] + n * 4].view(torch.float32)
    a["off"] += nb
    return out


def zero_arena_end() -> None:
    _ARENA["key"] = None
    _ARENA["ok"] = ()


def ws_float(name: str, n: int, device) -> Tensor:
    return workspace(name, n * 4, device)[: n * 4].view(torch.float32)


# ---------------------------------------------------------------------------------------------
# NHWC map helpers.  A "map" is a float32 tensor of shape [N,H,W,C] whose (H,W,C) dims are dense;
# the image stride may be larger (level views of a [N,A,C] buffer).
# ---------------------------------------------------------------------------------------------
def _check_map(t: Tensor) -> None:
    assert t.dtype in (torch.float32, torch.bfloat16) and t.dim() == 4, (t.dtype, t.shape)
    N, H, W, Cc = t.shape
    st = t.stride()
    assert st[3] == 1 and st[2] == Cc and st[1] == W * Cc, f"map is not NHWC-dense: {t.shape} {st}"


def level_views(cat: Tensor, sizes: Sequence[Tuple[int, int]]) -> List[Tensor]:
    """[N,A,C] level-concatenated buffer -> per-level [N,h,w,C] views (no copy)."""
    N, A, Cc = cat.shape
    out, off = [], 0
    for (h, w) in sizes:
        out.append(cat[:, off:off + h * w, :].unflatten(1, (h, w)))
        off += h * w
    assert off == A
    return out


def make_levels(sizes: Sequence[Tuple[int, int]]) -> Levels:
    lv = Levels()
    lv.nseg = len(sizes)
    off = 0
    for i, (h, w) in enumerate(sizes):
        lv.off[i] = off
        lv.cnt[i] = h * w
        off += h * w
    return lv


# ---------------------------------------------------------------------------------------------
# compute mode of the 1x1 / 3x3 convolutions: "f32" (fp32 matrix cores, the headline configuration), "bf16"
# (BASELINE.json configs[2]: both multiplicands rounded to bf16 on their way to the bf16 matrix cores, fp32
# accumulation, epilogues and storage) or "f32x3": fp32 maps / accumulation / results as in "f32", but the direct
# implicit-GEMM launches form every product on the bf16 matrix cores from exact three-limb splits of both fp32
# multiplicands (erd_conv_desc::w_x3; gfx950's fp32 MFMA runs at 1/16 of the bf16 rate), and so do the weight-gradient
# launches (erd_wgrad_desc::limbs3: both operands are split in the kernel's transposing loader) and the Winograd launches
# (erd_wino_conv3x3_x3: U = G g G^T pre-split, V = B^T d B split behind the transform; WINO_X3).  Process-wide switch:
# set_compute(...) / ERD_COMPUTE=...
# ---------------------------------------------------------------------------------------------
DEFAULT_COMPUTE = _os.environ.get("ERD_COMPUTE", "f32x3")      # the fp32 configuration of BASELINE configs[1]; "f32": native fp32 MFMA
COMPUTE = DEFAULT_COMPUTE


def set_compute(mode: str) -> None:
    global COMPUTE
    if mode not in ("f32", "bf16", "f32x3"):
        raise ValueError(f"compute mode {mode!r}: 'f32', 'f32x3' or 'bf16'")
    COMPUTE = mode


def act_dtype() -> torch.dtype:
    """storage type of feature maps and their gradients: bf16 in the bf16 mode (BASELINE.json configs[2]; the reference's
    AMP switch, tools/train.py:85-97), fp32 otherwise.  The image, the head outputs (gfl_head.py:224-229 feed the losses),
    every statistic / reduction / parameter / parameter gradient stay fp32 in both modes."""
    return torch.bfloat16 if COMPUTE == "bf16" and BF16_STORAGE else torch.float32


BF16_STORAGE = _os.environ.get("ERD_BF16_STORAGE", "1") != "0"    # 0: the round-1 form (bf16 multiplicands, fp32 maps): A/B aid


def _mt(*ts: Tensor) -> int:
    """erd_hip.h map_type of a group of maps that must share it (ERD_F32 = 0 | ERD_BF16 = 1)"""
    dts = {t.dtype for t in ts if t is not None}
    assert len(dts) == 1 and dts <= {torch.float32, torch.bfloat16}, f"maps must share one storage type: {dts}"
    return 1 if dts.pop() == torch.bfloat16 else 0


def to_bf16(t: Tensor) -> Tensor:
    assert t.is_contiguous() and t.dtype == torch.float32
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    call("erd_to_bf16", _p(t), _p(out), t.numel(), _stream())
    return out


def to_bf16_into(src: Tensor, dst: Tensor) -> None:
    assert src.is_contiguous() and dst.is_contiguous() and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    call("erd_to_bf16", _p(src), _p(dst), src.numel(), _stream())


def _weights_bf16(w: Tensor) -> Tensor:
    """bf16 copy of a contiguous OHWI weight view.  Frozen weights (teacher, stem/layer1) are converted once: the copy
    is cached ON the owning parameter object and validated by storage pointer + in-place version; trainable weights
    change under the optimizer's raw-pointer update, so they are converted per use (one small launch)."""
    base = getattr(w, "_erd_owner", None)
    if base is None:
        base = w._base if w._base is not None else w
    sh = getattr(base, "_erd_shadow", None)      # (bf16 view into the trainer's flat shadow buffer, version when refreshed)
    if sh is not None and sh[1] == base._version:
        v = sh[0].permute(0, 2, 3, 1)
        if v.shape == w.shape and v.is_contiguous():
            return v
    if base.requires_grad or w.requires_grad:
        return to_bf16(w.detach())
    ver = (w.data_ptr(), base._version, tuple(w.shape))
    hit = getattr(base, "_erd_bf16", None)
    if hit is None or hit[0] != ver:
        hit = (ver, to_bf16(w.detach()))
        base._erd_bf16 = hit
    return hit[1]


def split3(t: Tensor) -> Tensor:
    """[3, *t.shape] bf16: the three limb planes of a contiguous fp32 tensor (hi + mid + lo == t exactly)"""
    assert t.is_contiguous() and t.dtype == torch.float32
    out = torch.empty((3,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    call("erd_split3", _p(t), _p(out), t.numel(), _stream())
    return out


def _weights_x3(w: Tensor, grad_form: bool = False) -> Tensor:
    """limb planes of a contiguous weight view for erd_conv_desc::w_x3.  Forward form: `w` is the OHWI view of a parameter --
    frozen ones are split once (cached on the parameter, validated by pointer + version), trainable ones are served from the
    trainer's per-step preparation (ParamPrep, kind 3) or split per use.  Gradient form: `w` is a transposed (BN-scaled)
    weight from weight_transpose(); when that buffer is a prepared one its planes are prepared too (second launch)."""
    owner = getattr(w, "_erd_prep_owner", None) if grad_form else _prep_owner(w)
    prep = _prep_of(owner)
    key = ("XT" if grad_form else "X", id(owner))
    if prep is not None:
        r = prep.lookup(key)
        if r is not None and r.matches(owner, w, None):
            return r.out
    if not grad_form:
        base = getattr(w, "_erd_owner", None)
        if base is not None and not base.requires_grad and not w.requires_grad:
            ver = (w.data_ptr(), base._version, tuple(w.shape))
            hit = getattr(base, "_erd_x3", None)
            if hit is None or hit[0] != ver:
                hit = (ver, split3(w.detach()))
                base._erd_x3 = hit
            return hit[1]
    out = split3(w.detach())
    if prep is not None and not torch.cuda.is_current_stream_capturing():
        prep.register(key, 3, w, None, torch.empty_like(out), w.shape[0], w.shape[1] * w.shape[2], w.shape[3], 0, owner,
                      1 if grad_form else 0)
    return out


# ---------------------------------------------------------------------------------------------
# convolution (forward form, input-gradient form, weight gradient)
# ---------------------------------------------------------------------------------------------
def _fill_seg(sg, x: Tensor, out: Tensor, GH: int, GW: int, res: Optional[Tensor], alpha: Optional[Tensor],
              mask: Optional[Tensor] = None):
    _check_map(x)
    _check_map(out)
    sg.inp = x.data_ptr()
    sg.out = out.data_ptr()
    sg.res = 0 if res is None else res.data_ptr()
    sg.alpha = 0 if alpha is None else alpha.data_ptr()
    sg.mask = 0
    if mask is not None:
        _check_map(mask)
        assert mask.shape == out.shape and mask.stride(0) == out.stride(0), "mask must share the output geometry"
        sg.mask = mask.data_ptr()
    sg.N, sg.IH, sg.IW = x.shape[0], x.shape[1], x.shape[2]
    sg.GH, sg.GW = GH, GW
    sg.OH, sg.OW = out.shape[1], out.shape[2]
    sg.in_nstride = x.stride(0)
    sg.out_nstride = out.stride(0)
    if res is not None:
        _check_map(res)
        assert res.shape == out.shape and res.stride(0) == out.stride(0), "residual must share the output geometry"
        sg.res_nstride = res.stride(0)


def _stored_bf16(*groups) -> int:
    """1 when the maps of these groups are stored as bf16, 0 for fp32; one launch never mixes the two within a role
    (inputs | outputs + residuals + masks)."""
    dts = {t.dtype for g in groups if g is not None for t in g if t is not None}
    assert len(dts) == 1, f"maps of one role must share a storage type: {dts}"
    bf = dts.pop() == torch.bfloat16
    assert not bf or COMPUTE == "bf16", "bf16-stored maps need set_compute('bf16')"
    return 1 if bf else 0


STREAMK = _os.environ.get('ERD_STREAMK', '1') != '0'     # stream-K work decomposition of the implicit-GEMM launches (see conv_mfma.hip)
_SK_TILES = 1 << 16


# Stream-K in data-parallel runs.  Round 1 switched it off at world size > 1 (a static split over 2 x #CU workgroups: CUs
# taken by RCCL channels make some workgroups start late and hold their tiles' hand-over back).  Measured in round 2 with a
# dummy kernel holding k half-CUs for the whole launch (tools/bench_wgrad_sensitivity.py, 1024->256 1x1 on 4 x 50 x 84):
# stream-K 119 / 124 / 125 / 139 us at k = 0 / 4 / 16 / 32 against 162 us for the tile-parallel launch at every k -- the
# split degrades gracefully and stays ahead even with an eighth of the chip taken, so every world size runs the SAME kernel
# configuration (ERD_STREAMK_MULTIRANK=0 restores the old behaviour; the N = 1 point of a scaling curve is then NOT the
# sibling of the N > 1 points).
STREAMK_MULTIRANK = _os.environ.get("ERD_STREAMK_MULTIRANK", "1") != "0"


def _multi_rank() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


_SK_BYTES = None


def _attach_sk_ws(d: ConvDesc, device) -> None:
    # (one zero-initialised ticket / slab workspace per stream; see STREAMK_MULTIRANK above for the multi-rank policy)
    global _SK_BYTES
    if not STREAMK or (not STREAMK_MULTIRANK and _multi_rank()):
        d.sk_ws, d.sk_ws_bytes = 0, 0
        return
    if _SK_BYTES is None:
        _SK_BYTES = int(_lib.load().erd_conv_igemm_ws_bytes(_SK_TILES))
    nbytes = _SK_BYTES
    key = ("streamk", device, _cur_stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)    # tickets must start at zero
        _ws_replace(key, ws)
    d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), nbytes


# Launch descriptors are cached per call-site geometry: the host side of a step is ~250 convolution launches, and
# filling a ctypes erd_conv_desc field by field (taps, extents, strides, checks) cost ~25 us each -- more than the
# launch itself (measured: 6 of the 20 ms of host time per bf16 step, tools/host_breakdown.py).  A cached descriptor
# keeps everything that depends only on shapes / strides / storage types; a call rewrites the pointers and launches
# (the C side copies the struct into the kernel arguments before it returns).  One cache per host thread: the backward
# pass runs on autograd's thread.
import threading as _threading
_DESC = _threading.local()


def _desc_cache() -> dict:
    c = getattr(_DESC, "cache", None)
    if c is None:
        c = _DESC.cache = {}
    elif len(c) > 16384:      # (~60 input shapes of a mixed-resolution run; rebuilt on demand)
        c.clear()
    return c


def _igemm_class(d, form: str) -> str:
    """timing class of an erd_conv_igemm launch = the kernel SYMBOL it runs on (bench.py's roofline rows are per symbol, and must
    agree with rocprofv3's): the library's own dispatch predicate decides (erd_conv_thin_ok = erd::conv_thin_x3_ok,
    csrc/conv_thin.hip; only consulted while timing is on)"""
    if _TIMING is None:
        return "conv_igemm_" + form
    return ("conv_thin_" if _lib.load().erd_conv_thin_ok(C.byref(d)) else "conv_igemm_") + form


def _geom(ts) -> tuple:
    # (all strides: the NHWC-density check of _check_map runs when the descriptor is built, so a differently laid out
    # tensor of the same shape must not hit an entry that was checked for another layout)
    return tuple(None if t is None else (t.shape, t.stride(), t.dtype) for t in ts)


def conv_out_size(h: int, k: int, s: int, p: int) -> int:
    return (h + 2 * p - k) // s + 1


def conv_forward(xs: Sequence[Tensor], w: Tensor, outs: Sequence[Tensor], k: int, stride: int, pad: int,
                 scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                 res: Optional[Sequence[Optional[Tensor]]] = None, alphas: Optional[Sequence[Tensor]] = None,
                 relu: bool = False) -> None:
    """outs[i] = epi(conv(xs[i], w)); w is [Cout,k,k,Cin] contiguous (OHWI).  All segments share w."""
    _require_gpu(w, *xs, *outs)
    Cout, Cin = w.shape[0], w.shape[3]
    if alphas is None and wino_ok(Cin, k, stride, pad) and w.shape[1] == 3 and (WINO_TRAIN_FWD if RECORDED else WINO_NOGRAD_FWD):
        wino_conv3x3(xs, _wino_weights_cached(w), outs, Cout, scale=scale, shift=shift, relu=relu, res=res)
        return
    cache = _desc_cache()
    key = ("fwd", k, stride, pad, relu, w.shape, _geom(xs), _geom(outs), None if res is None else _geom(res), alphas is not None,
           scale is not None, shift is not None, COMPUTE)
    ent = cache.get(key)
    if ent is None:
        assert w.is_contiguous() and w.shape[1] == k and w.shape[2] == k
        d = ConvDesc()
        d.nseg = len(xs)
        for i, (x, o) in enumerate(zip(xs, outs)):
            assert x.shape[3] == Cin and o.shape[3] == Cout
            assert o.shape[1] == conv_out_size(x.shape[1], k, stride, pad) and o.shape[2] == conv_out_size(x.shape[2], k, stride, pad)
            _fill_seg(d.seg[i], x, o, o.shape[1], o.shape[2], None if res is None else res[i],
                      None if alphas is None else alphas[i])
        d.Cin, d.Cout, d.wrow = Cin, Cout, k * k * Cin
        d.ntaps = k * k
        for kh in range(k):
            for kw in range(k):
                t = kh * k + kw
                d.dy[t], d.dx[t], d.wk[t] = kh - pad, kw - pad, t * Cin
        d.in_stride, d.out_stride, d.oy, d.ox = stride, 1, 0, 0
        d.relu = 1 if relu else 0
        d.colsum = 0
        d.in_bf16, d.out_bf16 = _stored_bf16(xs), _stored_bf16(outs, res)
        flop = 2.0 * sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs) * Cout * k * k * Cin
        nbytes = 4.0 * (sum(x.numel() for x in xs) + sum(o.numel() for o in outs) + w.numel())
        ent = cache[key] = (d, flop, nbytes, _shape_tag(d, k, stride))
    d, flop, nbytes, tag = ent
    for i in range(len(xs)):
        sg = d.seg[i]
        sg.inp, sg.out = xs[i].data_ptr(), outs[i].data_ptr()
        if res is not None:
            sg.res = 0 if res[i] is None else res[i].data_ptr()
        if alphas is not None:
            sg.alpha = alphas[i].data_ptr()
    d.w = w.data_ptr()
    d.scale = 0 if scale is None else scale.data_ptr()
    d.shift = 0 if shift is None else shift.data_ptr()
    if COMPUTE == "bf16":
        wb = _weights_bf16(w)          # (kept alive by this frame until the launch is queued; stream-ordered free)
        d.w_bf16 = wb.data_ptr()
    elif COMPUTE == "f32x3" and Cin % 4 == 0:
        wx = _weights_x3(w)
        d.w_x3 = wx.data_ptr()
    else:
        d.w_x3 = 0
    _attach_sk_ws(d, w.device)
    _timed_call(_igemm_class(d, "fwd"), flop, "erd_conv_igemm", C.byref(d), _stream(), nbytes=nbytes if _TIMING is not None else 0.0,
                tag=tag if TIMING_DETAIL else "")


WINOGRAD = _os.environ.get("ERD_WINO", "1") != "0"     # F(2x2,3x3) for the fp32 3x3 stride-1 convolutions (winograd.hip)


WINO_X3 = _os.environ.get("ERD_WINO_X3", "1") != "0"      # "f32x3": the Winograd launches on the bf16 matrix cores too (A/B aid: 0)


def wino_x3() -> bool:
    """do the Winograd launches run in the three-limb form (erd_wino_conv3x3_x3)?"""
    return WINO_X3 and COMPUTE == "f32x3"


def wino_weights(w_ohwi: Tensor, flip: bool = False, x3: Optional[bool] = None) -> Tensor:
    """U = G g G^T of a [Cout,3,3,Cin] weight in the layout erd_wino_conv3x3 streams (flip: taps reversed, for the
    input-gradient form on transposed weights); x3 (default: wino_x3()): the bf16 limb image erd_wino_conv3x3_x3 streams"""
    Cout, kh, kw, Cin = w_ohwi.shape
    assert kh == 3 and kw == 3 and w_ohwi.is_contiguous() and w_ohwi.dtype == torch.float32
    x3 = wino_x3() if x3 is None else x3
    # forward form: the source is the parameter; gradient form: a prepared transposed weight
    owner = _prep_owner(w_ohwi) if not flip else getattr(w_ohwi, "_erd_prep_owner", None)
    PREP = _prep_of(owner)
    if PREP is None:
        owner = None
    key = (("UT" if flip else "U") + ("3" if x3 else ""), id(owner))
    if owner is not None:
        r = PREP.lookup(key)
        if r is not None and r.matches(owner, w_ohwi, None):
            return r.out
    if x3:
        U = torch.empty(int(_lib.load().erd_wino_weights_x3_elems(Cout, Cin)), dtype=torch.bfloat16, device=w_ohwi.device)
        call("erd_wino_weights_x3", _p(w_ohwi), _p(U), Cout, Cin, 1 if flip else 0, _stream())
    else:
        U = torch.empty(int(_lib.load().erd_wino_weights_elems(Cout, Cin)), dtype=torch.float32, device=w_ohwi.device)
        call("erd_wino_weights", _p(w_ohwi), _p(U), Cout, Cin, 1 if flip else 0, _stream())
    if owner is not None and not torch.cuda.is_current_stream_capturing():
        PREP.register(key, 4 if x3 else 2, w_ohwi, None, torch.empty_like(U), Cout, 9, Cin, 1 if flip else 0, owner, 1 if flip else 0)
    return U


def _wino_weights_cached(w: Tensor) -> Tensor:
    """frozen weights: transformed once (cached on the owning parameter, validated by pointer + version + form);
    trainable ones per use (a 5 us launch)."""
    base = getattr(w, "_erd_owner", None)
    if base is None or base.requires_grad:
        return wino_weights(w)
    x3 = wino_x3()
    ver = (w.data_ptr(), base._version, tuple(w.shape), x3)
    attr = "_erd_wino3" if x3 else "_erd_wino"
    hit = getattr(base, attr, None)
    if hit is None or hit[0] != ver:
        hit = (ver, wino_weights(w, x3=x3))
        setattr(base, attr, hit)
    return hit[1]


_WINO_SCHED = {}


def _wino_sched(device) -> Tensor:
    """two zero-initialised ints per stream: the item counter of the persistent Winograd grid (the kernel leaves them
    zero, launches on one stream are ordered)"""
    key = (str(device), _cur_stream())
    t = _WINO_SCHED.get(key)
    if t is None:
        t = _WINO_SCHED[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


# Where the Winograd kernels may run (all default on; `distillation_forward` below narrows the no-grad case): recorded forward passes (the student's), no-grad forward passes (inference), input
# gradients.  RECORDED is set by every fused op's forward: does autograd record this pass (any input needs a gradient)?
WINO_TRAIN_FWD = _os.environ.get("ERD_WINO_TRAIN_FWD", "1") != "0"
WINO_NOGRAD_FWD = _os.environ.get("ERD_WINO_NOGRAD_FWD", "1") != "0"
WINO_DGRAD = _os.environ.get("ERD_WINO_DGRAD", "1") != "0"
WINO_TEACHER = _os.environ.get("ERD_WINO_TEACHER", "1") != "0"
WINO_FROZEN_TRUNK = _os.environ.get("ERD_WINO_FROZEN_TRUNK", "0") == "1"
RECORDED = False


class distillation_forward:
    """Scope in which the convolutions autograd does NOT record (a frozen teacher; the student's frozen stem / layer1)
    may use the Winograd kernels or not.  Measured at full size on one input (tests/diag/diag_wino_matrix.py, all 16
    placements, gradients against the CPU oracle; DESIGN.md 3 has the seed sweep and the fp64 comparison): Winograd in the teacher and in the input gradients changes nothing (2.0e-4 with
    every other kernel direct), in the student's recorded layers 7.2e-4, in the student's FROZEN TRUNK 1.5e-3 -- although
    each of these layers is, taken alone, slightly closer to fp64 on Winograd than on the direct kernel
    (tests/diag/diag_layer1_error.py): a perturbation of the student's earliest activations is amplified through all
    trainable layers behind it.  Hence: teacher on Winograd (WINO_TEACHER), student trunk direct (WINO_FROZEN_TRUNK off)."""

    def __init__(self, winograd=None):
        self.winograd = WINO_FROZEN_TRUNK if winograd is None else winograd

    def __enter__(self):
        global WINO_NOGRAD_FWD
        self.keep = WINO_NOGRAD_FWD
        WINO_NOGRAD_FWD = self.winograd
        return self

    def __exit__(self, *exc):
        global WINO_NOGRAD_FWD
        WINO_NOGRAD_FWD = self.keep
        return False


def wino_ok(Cin: int, k: int, stride: int, pad: int) -> bool:
    # (16-channel K slices; the kernel's look-ahead pipeline wants at least four of them per item)
    return WINOGRAD and COMPUTE in ("f32", "f32x3") and k == 3 and stride == 1 and pad == 1 and Cin % 16 == 0 and Cin >= 64


def wino_conv3x3(xs: Sequence[Tensor], U: Tensor, outs: Sequence[Tensor], Cout: int, scale: Optional[Tensor] = None,
                 shift: Optional[Tensor] = None, relu: bool = False, res: Optional[Sequence[Optional[Tensor]]] = None,
                 mask: Optional[Sequence[Tensor]] = None, colsum: Optional[Tensor] = None, kname: str = "conv_wino_fwd") -> None:
    """outs[i] = epi(conv3x3 stride 1 pad 1 of xs[i]) by Winograd F(2x2,3x3); U = wino_weights(w)"""
    _require_gpu(U, *xs, *outs)
    from ._lib import ConvSeg
    segs = (ConvSeg * len(xs))()
    Cin = xs[0].shape[3]
    for i, (x, o) in enumerate(zip(xs, outs)):
        assert x.shape[:3] == o.shape[:3] and o.shape[3] == Cout and x.shape[3] == Cin
        assert x.dtype == torch.float32 and o.dtype == torch.float32, "the Winograd kernel is fp32 (maps and arithmetic)"
        _fill_seg(segs[i], x, o, x.shape[1], x.shape[2], None if res is None else res[i], None,
                  None if mask is None else mask[i])
    flop = 2.0 * sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs) * Cout * 9 * Cin
    nbytes = (4.0 * (sum(x.numel() for x in xs) + sum(o.numel() for o in outs)) + U.numel() * U.element_size()) if _TIMING is not None else 0.0
    # (the form is the weight image's: a bf16 image is the three-limb one).  Timed runs name the kernel SYMBOL the launch runs on:
    # erd_wino_conv3x3_x3 picks wino_x3p_kernel (128 couts per item) or wino_x3_kernel (64) by itself
    if _TIMING is not None and U.dtype == torch.bfloat16 and int(_lib.load().erd_wino_x3_couts_per_item(segs, len(xs), Cout)) == 128:
        kname += "_p"
    _timed_call(kname, flop, "erd_wino_conv3x3_x3" if U.dtype == torch.bfloat16 else "erd_wino_conv3x3", segs, len(xs), _p(U), Cin, Cout, _p(scale), _p(shift),
                1 if relu else 0, _p(colsum), (colsum.numel() // Cout if colsum is not None else 0),
                _p(_wino_sched(U.device)), _stream(), nbytes=nbytes,
                tag=f"px{sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs)} {Cin}->{Cout} k3s1" if TIMING_DETAIL else "")


# ---------------------------------------------------------------------------------------------
# per-step parameter preparation: everything the step derives from the convolution weights alone (transposed / BN-scaled
# weights of the input-gradient convolutions, Winograd weight images of both forms) is rebuilt by the trainer right after
# the optimizer update in two launches (erd_weight_prep_batch) instead of ~105 small launches scattered over the forward
# and backward passes.  The wrappers below serve a prepared buffer when one is valid for this step and otherwise launch
# as before -- and, under a trainer, leave a recipe so that the NEXT update prepares it.
# ---------------------------------------------------------------------------------------------
class _PrepRecipe:
    __slots__ = ("kind", "src", "rowscale", "out", "Cout", "ntaps", "Cin", "flip", "owner", "version", "level", "stamp",
                 "src_ptr", "rs_ptr")

    def matches(self, owner, src: Tensor, rowscale: Optional[Tensor]) -> bool:
        return (self.owner is owner and owner._version == self.version and src.data_ptr() == self.src_ptr and
                (0 if rowscale is None else rowscale.data_ptr()) == self.rs_ptr)


class ParamPrep:
    def __init__(self, device):
        self.device = torch.device(device)
        self.recipes: Dict[tuple, _PrepRecipe] = {}
        self.stamp = 0
        self._tables = None
        self._gtables: Dict[int, list] = {}
        self.group_of = None             # callable owner parameter -> group (run_group)
        self.on_stale = None             # callable: prepared buffers stopped vouching for the parameters (captured step graphs
                                         # read them by address: the trainer drops those graphs)

    def _stale(self) -> None:
        if self.on_stale is not None:
            self.on_stale()

    def lookup(self, key) -> Optional[_PrepRecipe]:
        r = self.recipes.get(key)
        return r if r is not None and r.stamp == self.stamp and self.stamp > 0 else None

    def register(self, key, kind: int, src: Tensor, rowscale: Optional[Tensor], out: Tensor, Cout: int, ntaps: int, Cin: int,
                 flip: int, owner, level: int) -> None:
        r = _PrepRecipe()
        r.kind, r.src, r.rowscale, r.out, r.Cout, r.ntaps, r.Cin, r.flip = kind, src, rowscale, out, Cout, ntaps, Cin, flip
        r.owner, r.version, r.level, r.stamp = owner, owner._version, level, -1
        r.src_ptr, r.rs_ptr = src.data_ptr(), 0 if rowscale is None else rowscale.data_ptr()
        out._erd_prep_owner = owner
        if key[0] == "T":                       # what was built from the previous transposed buffer is orphaned
            self.recipes.pop(("UT", key[1]), None)
            self.recipes.pop(("UT3", key[1]), None)
            self.recipes.pop(("XT", key[1]), None)
        self.recipes[key] = r
        self._tables = None
        self._gtables = {}

    def invalidate(self) -> None:
        self.stamp += 1
        self._stale()

    def _drop_stale(self, items) -> None:
        for key, r in items:      # re-homed storage / edited parameters: the recipe is stale
            if r.src.data_ptr() != r.src_ptr or (r.rowscale is not None and r.rowscale.data_ptr() != r.rs_ptr) or \
                    r.owner._version != r.version:
                self.recipes.pop(key, None)
                if key[0] == "T":
                    self.recipes.pop(("UT", key[1]), None)
                    self.recipes.pop(("UT3", key[1]), None)
                    self.recipes.pop(("XT", key[1]), None)
                self._tables = None
                self._gtables = {}
                self._stale()

    def _build_tables(self, recipes) -> list:
        from ._lib import WeightPrepItem
        lib = _lib.load()
        tables = []
        for level in (0, 1):
            rs = [r for r in recipes if r.level == level]
            if not rs:
                continue
            items = (WeightPrepItem * len(rs))()
            blk = 0
            for it, r in zip(items, rs):
                it.w, it.rowscale, it.dst = r.src_ptr, r.rs_ptr, r.out.data_ptr()
                it.Cout, it.ntaps, it.Cin, it.flip, it.kind, it.block0 = r.Cout, r.ntaps, r.Cin, r.flip, r.kind, blk
                blk += int(lib.erd_weight_prep_blocks(r.kind, r.Cout, r.ntaps, r.Cin))
            table = torch.frombuffer(bytearray(bytes(memoryview(items))), dtype=torch.uint8).to(self.device)
            tables.append((table, len(rs), blk, rs))
        return tables

    def run_group(self, g: int) -> None:
        """run() for the recipes of ONE group of parameters (`group_of(owner) == g`; ERDTrainer: a gradient bucket), on the
        current stream, right after that group's optimizer update and BN fold.  The other groups' buffers keep vouching for
        their parameters (no stamp change): a trainer that updates per group must run every group once per step."""
        if self.group_of is None:
            raise RuntimeError("ParamPrep.run_group: no group_of")
        if self.stamp == 0:
            self.stamp = 1
        mine = [(k, r) for k, r in self.recipes.items() if self.group_of(r.owner) == g]
        self._drop_stale(mine)
        tables = self._gtables.get(g)
        if tables is None:
            tables = self._gtables[g] = self._build_tables([r for r in self.recipes.values() if self.group_of(r.owner) == g])
        for table, n, blocks, rs in tables:
            call("erd_weight_prep_batch", _p(table), n, blocks, _stream())
            for r in rs:
                r.stamp = self.stamp

    def run(self) -> None:
        """rebuild every registered buffer from the current parameters (call right after the optimizer update, on the
        stream the step runs on, after the batched BN fold whose results the row scales are)"""
        self.stamp += 1
        if not self.recipes:
            return
        self._drop_stale(list(self.recipes.items()))
        if self._tables is None:
            self._tables = self._build_tables(list(self.recipes.values()))
        for table, n, blocks, rs in self._tables:
            call("erd_weight_prep_batch", _p(table), n, blocks, _stream())
            for r in rs:
                r.stamp = self.stamp


def _prep_owner(w: Tensor):
    o = getattr(w, "_erd_owner", None)
    return o if o is not None and getattr(o, "_erd_sink", False) else None


def _prep_of(owner) -> Optional[ParamPrep]:
    """the ParamPrep of the trainer that owns this parameter (ERDTrainer tags its flat parameters with `_erd_prep`):
    two trainers in one process (a test suite; a train + fine-tune pair) each serve their own prepared buffers"""
    return None if owner is None else getattr(owner, "_erd_prep", None)


def weight_transpose(w: Tensor, rowscale: Optional[Tensor] = None) -> Tensor:
    """[Cout,k,k,Cin] -> [Cin,k,k,Cout] (* rowscale[co]): weights of the input-gradient convolution."""
    Cout, k, _, Cin = w.shape
    bf = COMPUTE == "bf16"
    owner = _prep_owner(w)
    PREP = _prep_of(owner)
    if PREP is None:
        owner = None
    if owner is not None:
        r = PREP.lookup(("T", id(owner), bf))
        if r is not None and r.matches(owner, w, rowscale):
            return r.out
    if bf:      # rounded on the way out: conv_dgrad hands it to the bf16 matrix cores as is
        wt = torch.empty((Cin, k, k, Cout), dtype=torch.bfloat16, device=w.device)
        call("erd_weight_transpose_bf16", _p(w), _p(rowscale), _p(wt), Cout, k * k, Cin, 0, _stream())
    else:
        wt = torch.empty((Cin, k, k, Cout), dtype=torch.float32, device=w.device)
        call("erd_weight_transpose", _p(w), _p(rowscale), _p(wt), Cout, k * k, Cin, 0, _stream())
    if owner is not None and (rowscale is None or getattr(rowscale, "_erd_stable", False)) and not torch.cuda.is_current_stream_capturing():
        # (a row scale that is not a view of the trainer's batched BN fold changes its address every step: not preparable)
        PREP.register(("T", id(owner), bf), 1 if bf else 0, w, rowscale, torch.empty_like(wt), Cout, k * k, Cin, 0, owner, 0)
    return wt


def conv_dgrad(dzs: Sequence[Tensor], wt: Tensor, dxs: Sequence[Tensor], k: int, stride: int, pad: int,
               accumulate: bool = False, res: Optional[Sequence[Tensor]] = None,
               relu_mask: Optional[Sequence[Tensor]] = None, colsum: Optional[Tensor] = None) -> None:
    """dxs[i] (+)= conv_transpose(dzs[i]); wt = weight_transpose(w) is [Cin,k,k,Cout].
    colsum: [Cin] or a replicated [copies, Cin] accumulator (copies a power of two; bn_dgamma folds the rows).
    stride 1: one launch; stride 2: one launch per output-parity class (no zero-multiplies).
    Pixels of dx that no tap reaches (k=1, stride 2) are NOT written: pass accumulate=True on a
    buffer that already holds the other branch's gradient, or zero it first."""
    _require_gpu(wt, *dzs, *dxs)
    Cin, Cout = wt.shape[0], wt.shape[3]       # of the forward conv
    if wt.dtype == torch.float32 and WINO_DGRAD and wino_ok(Cout, k, stride, pad):
        # the input gradient of a 3x3 stride-1 conv is a 3x3 stride-1 conv of dz with the flipped transposed weights
        wino_conv3x3(dzs, wino_weights(wt, flip=True), dxs, Cin, res=(dxs if accumulate else res), mask=relu_mask,
                     colsum=colsum, kname="conv_wino_dgrad")
        return
    wtb = None
    wtx = None
    if COMPUTE == "bf16":
        wtb = wt if wt.dtype == torch.bfloat16 else to_bf16(wt)
    elif wt.dtype != torch.float32:
        raise ValueError("conv_dgrad: bf16 weights in f32 compute mode")
    elif COMPUTE == "f32x3" and Cout % 4 == 0:
        wtx = _weights_x3(wt, grad_form=True)
    if stride == 2 and k == 3 and len(dzs) == 1 and MERGE_PARITY:
        _dgrad_s2_merged(dzs[0], wt, wtb, dxs[0], pad, Cin, Cout, accumulate, res, relu_mask, colsum, wtx=wtx)
        return
    cache = _desc_cache()
    key = ("dgrad", k, stride, pad, accumulate, wt.shape, _geom(dzs), _geom(dxs), None if res is None else _geom(res),
           None if relu_mask is None else _geom(relu_mask), None if colsum is None else colsum.numel(), COMPUTE)
    ent = cache.get(key)
    if ent is None:
        ent = cache[key] = _dgrad_descs(dzs, wt, dxs, k, stride, pad, accumulate, res, relu_mask, colsum, Cin, Cout)
    for d, flop, nbytes, tag in ent:
        for i in range(len(dzs)):
            sg = d.seg[i]
            sg.inp, sg.out = dzs[i].data_ptr(), dxs[i].data_ptr()
            if accumulate:
                sg.res = dxs[i].data_ptr()
            elif res is not None:
                sg.res = res[i].data_ptr()
            if relu_mask is not None:
                sg.mask = relu_mask[i].data_ptr()
        d.w = wt.data_ptr() if wtb is None else 0
        d.colsum = 0 if colsum is None else colsum.data_ptr()
        if wtb is not None:
            d.w_bf16 = wtb.data_ptr()
        d.w_x3 = 0 if wtx is None else wtx.data_ptr()
        _attach_sk_ws(d, wt.device)
        _timed_call(_igemm_class(d, "dgrad"), flop, "erd_conv_igemm", C.byref(d), _stream(), nbytes=nbytes if _TIMING is not None else 0.0,
                    tag=tag if TIMING_DETAIL else "")


def _dgrad_descs(dzs, wt, dxs, k, stride, pad, accumulate, res, relu_mask, colsum, Cin, Cout):
    """the launch descriptors (one per output-parity class) of an input gradient, pointers left to the caller"""
    out = []
    classes = [(0, 0)] if stride == 1 else [(py, px) for py in range(stride) for px in range(stride)]
    for (py, px) in classes:
        taps = []
        for kh in range(k):
            if (py + pad - kh) % stride:
                continue
            for kw in range(k):
                if (px + pad - kw) % stride:
                    continue
                taps.append(((py + pad - kh) // stride, (px + pad - kw) // stride, (kh * k + kw) * Cout))
        if not taps:
            continue
        d = ConvDesc()
        d.nseg = len(dzs)
        skip = False
        for i, (dz, dx) in enumerate(zip(dzs, dxs)):
            assert dz.shape[3] == Cout and dx.shape[3] == Cin
            GH = (dx.shape[1] - py + stride - 1) // stride
            GW = (dx.shape[2] - px + stride - 1) // stride
            if GH <= 0 or GW <= 0:
                skip = True
            r = dx if accumulate else (None if res is None else res[i])
            _fill_seg(d.seg[i], dz, dx, max(GH, 0), max(GW, 0), r, None, None if relu_mask is None else relu_mask[i])
        if skip and len(dzs) == 1:
            continue
        d.Cin, d.Cout, d.wrow = Cout, Cin, k * k * Cout     # roles swap: contraction over Cout
        d.ntaps = len(taps)
        for t, (dy, dx_, wk) in enumerate(taps):
            d.dy[t], d.dx[t], d.wk[t] = dy, dx_, wk
        d.in_stride, d.out_stride, d.oy, d.ox = 1, stride, py, px
        d.scale = 0
        d.shift = 0
        d.relu = 0
        d.colsum_copies = 0 if colsum is None else colsum.numel() // Cin        # [copies, Cin of the forward conv]
        d.in_bf16, d.out_bf16 = _stored_bf16(dzs), _stored_bf16(dxs, res, relu_mask)
        flop = 2.0 * sum(d.seg[i].N * d.seg[i].GH * d.seg[i].GW for i in range(d.nseg)) * Cin * len(taps) * Cout
        # (stride 2: each parity class reads dz once and writes a quarter of dx)
        nbytes = 4.0 * (sum(t.numel() for t in dzs) + sum(t.numel() for t in dxs) / (stride * stride) + wt.numel() /
                        (stride * stride))
        out.append((d, flop, nbytes, _shape_tag(d, k, stride) + f" class{py}{px}"))
    return out


MERGE_PARITY = _os.environ.get("ERD_MERGE_PARITY", "1") != "0"


def _dgrad_s2_merged(dz: Tensor, wt: Tensor, wtb, dx: Tensor, pad: int, Cin: int, Cout: int, accumulate: bool, res,
                     relu_mask, colsum, wtx=None) -> None:
    """The four output-parity classes of a 3x3 / stride-2 input gradient (4 + 2 + 2 + 1 taps) as four segments of ONE
    launch with per-segment tap sets: ~1000 tiles of mixed length in one grid instead of four ragged launches of ~260."""
    k, stride = 3, 2
    d = ConvDesc()
    nseg, tap0, flop = 0, 0, 0.0
    for (py, px) in ((1, 1), (0, 1), (1, 0), (0, 0)):              # longest K loops first
        taps = [((py + pad - kh) // stride, (px + pad - kw) // stride, (kh * k + kw) * Cout)
                for kh in range(k) if (py + pad - kh) % stride == 0
                for kw in range(k) if (px + pad - kw) % stride == 0]
        GH = (dx.shape[1] - py + stride - 1) // stride
        GW = (dx.shape[2] - px + stride - 1) // stride
        if not taps or GH <= 0 or GW <= 0:
            continue
        sg = d.seg[nseg]
        r = dx if accumulate else (None if res is None else res[0])
        _fill_seg(sg, dz, dx, GH, GW, r, None, None if relu_mask is None else relu_mask[0])
        sg.tap0, sg.ntaps, sg.oy, sg.ox = tap0, len(taps), py, px
        for t, (dy, dx_, wk) in enumerate(taps):
            d.dy[tap0 + t], d.dx[tap0 + t], d.wk[tap0 + t] = dy, dx_, wk
        tap0 += len(taps)
        flop += 2.0 * sg.N * GH * GW * Cin * len(taps) * Cout
        nseg += 1
    if nseg == 0:
        return
    d.nseg = nseg
    d.ntaps = max(d.seg[i].ntaps for i in range(nseg))
    d.w = wt.data_ptr() if wtb is None else 0
    d.Cin, d.Cout, d.wrow = Cout, Cin, k * k * Cout     # roles swap: contraction over Cout
    d.in_stride, d.out_stride, d.oy, d.ox = 1, stride, 0, 0
    d.scale = d.shift = 0
    d.relu = 0
    d.colsum = 0 if colsum is None else colsum.data_ptr()
    d.colsum_copies = 0 if colsum is None else colsum.numel() // Cin
    if wtb is not None:
        d.w_bf16 = wtb.data_ptr()
    d.w_x3 = 0 if wtx is None else wtx.data_ptr()
    d.in_bf16, d.out_bf16 = _stored_bf16([dz]), _stored_bf16([dx], res, relu_mask)
    d.sk_ws, d.sk_ws_bytes = 0, 0
    nbytes = 4.0 * (dz.numel() + dx.numel() + wt.numel()) if _TIMING is not None else 0.0
    _timed_call("conv_igemm_dgrad", flop, "erd_conv_igemm", C.byref(d), _stream(), nbytes=nbytes,
                tag=(_shape_tag(d, k, stride) + " merged") if TIMING_DETAIL else "")


def _shape_tag(d, k: int, stride: int) -> str:
    px = sum(d.seg[i].N * d.seg[i].GH * d.seg[i].GW for i in range(d.nseg))
    return f"px{px} {d.Cin}->{d.Cout} k{k}s{stride}"


WGRAD_X3_GENERIC = _os.environ.get("ERD_WGRAD_X3_GENERIC", "1") != "0"      # ... and of the 1x1 / stride-2 layers
WGRAD_X3 = _os.environ.get("ERD_WGRAD_X3", "1") != "0"      # three-limb form of the three-tap weight gradient in the f32x3 mode (A/B aid)


# CUs left to somebody else's resident kernels (RCCL under data parallelism): the library sizes its persistent / stream-K grids for the
# rest (erd_set_cu_reserve), the one-round weight-gradient splits below follow (erd_usable_cus).  0 = the whole chip (every N = 1 run).
CU_RESERVE = 0


def set_cu_reserve(n: int) -> int:
    """returns the previous reserve; cached launch descriptors are keyed on it"""
    global CU_RESERVE
    prev = int(_lib.load().erd_set_cu_reserve(int(n)))
    CU_RESERVE = int(n)
    return prev


def _one_round(target: int) -> int:
    """a split target that means `k workgroups per CU on every CU` (768 = 3 x 256, ...), scaled to the CUs in use"""
    if CU_RESERVE == 0:
        return target
    usable = int(_lib.load().erd_usable_cus())
    return max(1, target * usable // (usable + CU_RESERVE))


def _pick_nsplit(npix: int, Cout: int, Cin: int, ntaps: int) -> int:
    tiles = ((Cout + 127) // 128) * ((Cin + 127) // 128) * ntaps
    kt = (npix + 31) // 32
    # ONE whole dispatch round: 4 workgroups per CU for the fp32 kernel (1024), TWO for the bf16 mode's kernel (conv_wgrad_bf16_kernel,
    # launch bounds 2: 512 -- until round 6 it got 1024 = two rounds and twice the partial slabs, whose fp32 traffic exceeds the bf16
    # operands': 229 -> 240 img/s in the bf16 mode, sweep 384 / 512 / 640 / 768 / 1024 in profiles/r06_bf16_wgrad_split.txt)
    dflt = "512" if COMPUTE == "bf16" else "1024"
    target = _one_round(int(dflt) if _os.environ.get("ERD_WGRAD_VARIANT", "1") == "0" else int(_os.environ.get("ERD_WGRAD_TARGET", dflt)))
    want = max(1, target // tiles)          # (measured: two rounds pay more partial-slab traffic than they gain; never a ragged extra round)
    return int(max(1, min(want, kt // 8 if kt >= 8 else 1, 512)))


def _extent(t: Tensor):
    """(base data_ptr, element count) of the allocation slice a strided map spans"""
    last = sum((sz - 1) * st for sz, st in zip(t.shape, t.stride()))
    return t.data_ptr(), last + 1


def conv_wgrad_partials(xs: Sequence[Tensor], dzs: Sequence[Tensor], k: int, stride: int, pad: int):
    """returns (part [S, Cout, k*k, Cin], S): split-K partial slabs.  All segments (maps sharing the weights, e.g.
    the head's five levels) are summed in ONE launch: their pixels are concatenated along the GEMM K axis."""
    _require_gpu(*xs, *dzs)
    xb = min(x.data_ptr() for x in xs)
    zb = min(z.data_ptr() for z in dzs)
    xoff = tuple((x.data_ptr() - xb) // x.element_size() for x in xs)
    zoff = tuple((z.data_ptr() - zb) // z.element_size() for z in dzs)
    cache = _desc_cache()
    key = ("wgrad", k, stride, pad, _geom(xs), _geom(dzs), xoff, zoff, COMPUTE, _os.environ.get("ERD_WGRAD_ROW3", "1"), WGRAD_X3, WGRAD_X3_GENERIC,
           CU_RESERVE)
    ent = cache.get(key)
    if ent is None:
        ent = cache[key] = _wgrad_desc(xs, dzs, k, stride, pad, xoff, zoff)
    d, S, flop, nbytes, tag, Cout, Cin, row3 = ent
    d.x, d.dz = xb, zb
    part = ws_float("wgrad_part", S * Cout * k * k * Cin, xs[0].device)
    d.part = part.data_ptr()
    # (timing classes follow the kernel SYMBOL: the three-tap kernel of the 3x3 / stride-1 layers vs the generic one)
    _timed_call("conv_wgrad_row3" if row3 else "conv_wgrad", flop, "erd_conv_wgrad", C.byref(d), _stream(), nbytes=nbytes if _TIMING is not None else 0.0,
                tag=tag if TIMING_DETAIL else "")
    return part, S


def _wgrad_desc(xs, dzs, k, stride, pad, xoff, zoff):
    """the launch descriptor of a weight gradient (base pointers and the partial-slab pointer left to the caller)"""
    Cin, Cout = xs[0].shape[3], dzs[0].shape[3]
    npix = sum(dz.shape[0] * dz.shape[1] * dz.shape[2] for dz in dzs)
    d = WgradDesc()
    xe = ze = 0
    d.nseg = len(xs)
    for i, (x, dz) in enumerate(zip(xs, dzs)):
        _check_map(x)
        _check_map(dz)
        sg = d.seg[i]
        _, nx = _extent(x)
        _, nz = _extent(dz)
        sg.x_off, sg.dz_off = xoff[i], zoff[i]
        xe, ze = max(xe, sg.x_off + nx), max(ze, sg.dz_off + nz)
        sg.N, sg.IH, sg.IW = x.shape[0], x.shape[1], x.shape[2]
        sg.GH, sg.GW, sg.OH, sg.OW = dz.shape[1], dz.shape[2], dz.shape[1], dz.shape[2]
        sg.x_nstride, sg.dz_nstride = x.stride(0), dz.stride(0)
    d.x_elems, d.dz_elems = xe, ze
    d.Cin, d.Cout, d.ntaps = Cin, Cout, k * k
    for kh in range(k):
        for kw in range(k):
            d.dy[kh * k + kw], d.dx[kh * k + kw] = kh - pad, kw - pad
    d.in_stride, d.out_stride, d.oy, d.ox = stride, 1, 0, 0
    d.bf16_multiplicands = 1 if COMPUTE == "bf16" else 0
    d.x_bf16, d.dz_bf16 = _stored_bf16(xs), _stored_bf16(dzs)
    row3 = int(_lib.load().erd_wgrad_row3_slices(C.byref(d))) if _os.environ.get("ERD_WGRAD_ROW3", "1") != "0" else 0
    d.limbs3 = 1 if (COMPUTE == "f32x3" and WGRAD_X3 and (row3 or WGRAD_X3_GENERIC)) else 0
    if row3 and d.limbs3:      # three-limb form: (128 | 64) output channels x 64 input channels x 3 taps per workgroup, one dispatch round
        bmr = 128 if Cout > 64 else 64
        groups = ((Cout + bmr - 1) // bmr) * ((Cin + 63) // 64) * 3
        target = _one_round(int(_os.environ.get("ERD_WGRAD_ROW3_X3_TARGET", "768")))      # three workgroups per CU: one dispatch round
        S = int(max(1, min(target // groups, row3 // 16 if row3 >= 16 else 1, 512)))
    elif row3:      # three taps per workgroup, two workgroups per CU: ONE whole dispatch round of (cout, cin, ky, split) workgroups
        # (measured best: 2 or 3 rounds pay more partial-slab traffic than they gain; a ragged extra round costs 15-40 %)
        bme = 64 if Cout <= 64 else (96 if Cout <= 96 else 128)        # rows of the kernel's output tile (erd_conv_wgrad)
        groups = ((Cout + bme - 1) // bme) * ((Cin + 127) // 128) * 3
        target = _one_round(int(_os.environ.get("ERD_WGRAD_ROW3_TARGET", "512")))
        S = int(max(1, min(target // groups, row3 // 16 if row3 >= 16 else 1, 512)))     # floor: whole dispatch rounds
    elif d.limbs3:      # three-limb form of the other layers: 128 x 128 channels, one tap per workgroup, one dispatch round of three per CU
        tiles = ((Cout + 127) // 128) * ((Cin + 127) // 128) * k * k
        kt = (npix + 15) // 16
        S = int(max(1, min(_one_round(int(_os.environ.get("ERD_WGRAD_X3_TARGET", "768"))) // tiles, kt // 8 if kt >= 8 else 1, 512)))
    else:
        S = _pick_nsplit(npix, Cout, Cin, k * k)
    d.nsplit = S
    flop = 2.0 * npix * Cout * Cin * k * k
    nbytes = 4.0 * (sum(t.numel() for t in xs) + sum(t.numel() for t in dzs) + S * Cout * k * k * Cin)
    return d, S, flop, nbytes, f"px{npix} {Cin}->{Cout} k{k}s{stride} S{S}", Cout, Cin, bool(row3)


def wgrad_reduce(part: Tensor, S: int, w: Tensor, rowscale: Optional[Tensor], dW: Tensor, accumulate: bool,
                 rowdot: Optional[Tensor], rowdot_zeroed: bool = False) -> None:
    """rowdot_zeroed: the caller took rowdot from the step's zero arena (zeros_f32): no memset launch"""
    Cout = w.shape[0]
    K = w.numel() // Cout
    call("erd_wgrad_reduce", _p(part), S, Cout, K, _p(w), _p(rowscale), _p(dW),
         (1 if accumulate else 0) | (2 if rowdot_zeroed else 0), _p(rowdot), _stream())


# ---------------------------------------------------------------------------------------------
# stem / pooling / norm helpers
# ---------------------------------------------------------------------------------------------
def stem(x_nchw: Tensor, w_ohwi: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    _require_gpu(x_nchw, w_ohwi)
    assert x_nchw.is_contiguous() and x_nchw.shape[1] == 3 and w_ohwi.shape == (64, 7, 7, 3) and w_ohwi.is_contiguous()
    N, _, H, W = x_nchw.shape
    OH, OW = conv_out_size(H, 7, 2, 3), conv_out_size(W, 7, 2, 3)
    y = torch.empty((N, OH, OW, 64), dtype=torch.float32, device=x_nchw.device)      # (stem output: fp32 scratch, pooled once)
    call("erd_stem_conv7x7_bn_relu", _p(x_nchw), _p(w_ohwi), _p(scale), _p(shift), _p(y), N, H, W, _stream())
    PH, PW = conv_out_size(OH, 3, 2, 1), conv_out_size(OW, 3, 2, 1)
    z = torch.empty((N, PH, PW, 64), dtype=act_dtype(), device=x_nchw.device)
    call("erd_maxpool3x3s2", _p(y), _p(z), N, OH, OW, 64, _mt(z), _stream())
    return z


def preprocess_into(img: Tensor, out_slot: Tensor, mean, std, flip_channels: bool, pad_value: float) -> None:
    """one CHW image (uint8 / fp32, on the GPU) -> out_slot [3,H,W] (a slice of the batch tensor)"""
    _require_gpu(img, out_slot)
    assert img.dim() == 3 and img.shape[0] == 3 and img.is_contiguous() and out_slot.is_contiguous()
    assert img.dtype in (torch.uint8, torch.float32), img.dtype
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    call("erd_preprocess_image", _p(img), int(img.dtype == torch.uint8), img.shape[1], img.shape[2], _p(out_slot),
         out_slot.shape[1], out_slot.shape[2], m, s, int(flip_channels), float(pad_value), _stream())


def resize_normalize_into(src_hwc_u8: Tensor, tables, new_hw, out_slot: Tensor, mean, std, flip: bool, swap_rb: bool,
                          pad_value: float) -> None:
    """decoded uint8 [h,w,3] image on the GPU -> out_slot [3,H,W]: bilinear resize to new_hw (fixed-point, the tables
    come from datasets.linear_coeffs), optional horizontal flip / channel swap, (v-mean)/std, pad_value elsewhere."""
    _require_gpu(src_hwc_u8, out_slot)
    assert src_hwc_u8.dtype == torch.uint8 and src_hwc_u8.dim() == 3 and src_hwc_u8.shape[2] == 3 and src_hwc_u8.is_contiguous()
    xo, xc, yo, yc = tables
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    call("erd_resize_normalize", _p(src_hwc_u8), src_hwc_u8.shape[0], src_hwc_u8.shape[1], _p(xo), _p(xc), _p(yo), _p(yc),
         int(new_hw[0]), int(new_hw[1]), _p(out_slot), out_slot.shape[1], out_slot.shape[2], m, sd, int(flip), int(swap_rb),
         float(pad_value), _stream())


def bn_fold(gamma: Tensor, beta: Tensor, mean: Tensor, var: Tensor, eps: float = 1e-5):
    _require_gpu(gamma)
    scale = torch.empty_like(gamma)
    shift = torch.empty_like(gamma)
    call("erd_bn_fold", _p(gamma), _p(beta), _p(mean), _p(var), eps, _p(scale), _p(shift), gamma.numel(), _stream())
    return scale, shift


def relu_bwd_colsum(y: Optional[Tensor], dy: Tensor, use_relu: bool, want_colsum: bool = True,
                    colsum_into: Optional[Tensor] = None):
    """(dz, colsum): dz = dy*(y>0) (new buffer) or dy itself; colsum[c] = sum over pixels of dz (accumulated into
    `colsum_into` when given -- it must hold zeros or a partial sum of the same quantity)."""
    _check_map(dy)
    N, H, W, Cc = dy.shape
    colsum = colsum_into if colsum_into is not None else (zeros_f32(Cc, dy.device) if want_colsum else None)
    dz = dy
    if use_relu:
        _check_map(y)
        assert y.is_contiguous() and dy.is_contiguous(), "relu backward runs on dense maps"
        dz = torch.empty_like(dy)
    call("erd_relu_bwd_colsum", _p(y), _p(dy), _p(dz), N * H * W, Cc, dy.stride(0), H * W, _p(colsum),
         1 if use_relu else 0, _mt(dy, y if use_relu else None), _stream())
    return dz, colsum


# Rows of a replicated column-sum accumulator (a power of two; erd_conv_desc::colsum_copies: workgroup b adds into row b mod copies).
# Eight rows were enough for the 1x1 input gradients of wide layers, not for the narrow ones: 525 persistent items x 4 waves adding to
# 8 x 128 addresses made the layer-2 Winograd input gradient 2x its forward twin (173 -> 103 us with 64 rows, 119 -> 99 for layer 3;
# tools/dbg/wino_dgrad_epi.py), and atomics share the in-order memory counter with the next item's loads.  So: as many rows as keep the
# accumulator at COLSUM_WIDTH floats (C = 64 / 128: 128 rows ... 2048: 8); bn_dgamma folds them, eight row lanes per channel.
COLSUM_COPIES = 8
COLSUM_WIDTH = int(_os.environ.get("ERD_COLSUM_WIDTH", "16384"))      # (0: eight rows everywhere, the rounds 2-5 layout)


def colsum_copies(channels: int) -> int:
    c = COLSUM_COPIES
    while COLSUM_WIDTH > 0 and c * 2 * channels <= COLSUM_WIDTH and c < 128:
        c *= 2
    return c


def bn_dgamma(rowdot: Tensor, dbeta: Tensor, mean: Tensor, var: Tensor, eps: float = 1e-5,
              out: Optional[Tensor] = None, dbeta_out: Optional[Tensor] = None) -> Tensor:
    """dbeta: [C] or the replicated [copies, C] accumulator of a gradient convolution's epilogue; its row sum is
    stored to `dbeta_out` when given"""
    Cc = dbeta.shape[-1]
    copies = dbeta.numel() // Cc
    dg = torch.empty(Cc, dtype=torch.float32, device=dbeta.device) if out is None else out
    call("erd_bn_dgamma", _p(rowdot), _p(dbeta), copies, _p(mean), _p(var), eps, _p(dg), _p(dbeta_out), 0, Cc, _stream())
    return dg


def colsum(x2d: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """x2d: contiguous [rows, C] -> [C] (stored into `out` when given)"""
    assert x2d.is_contiguous()
    rows, Cc = x2d.shape[:-1].numel(), x2d.shape[-1]
    if out is None:
        out = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    call("erd_colsum", _p(x2d), rows, Cc, _p(out), 0, _mt(x2d), _stream())
    return out


def gn_relu_forward(c: Tensor, gamma: Tensor, beta: Tensor, sizes, G: int = 32, eps: float = 1e-5):
    """c: [N,A,C] contiguous conv output -> (y, mean_rstd)"""
    _require_gpu(c)
    assert c.is_contiguous()
    N, A, Cc = c.shape
    lv = make_levels(sizes)
    y = torch.empty_like(c)
    nst = N * lv.nseg * G
    stats = workspace("gn_stats", nst * 16, c.device)
    mr = torch.empty((N, lv.nseg, G, 2), dtype=torch.float32, device=c.device)
    call("erd_gn_relu_fwd", _p(c), _p(y), _p(gamma), _p(beta), _p(stats), _p(mr), N, A, Cc, G, C.byref(lv), eps,
         _mt(c), _stream())
    return y, mr


GN_FUSED = _os.environ.get("ERD_GN_FUSED", "1") != "0"      # GroupNorm statistics from the producing convolution (A/B aid: 0)


def conv3x3_gn_relu_forward(x_cat: Tensor, w: Tensor, gamma: Tensor, beta: Tensor, sizes, G: int = 32, eps: float = 1e-5):
    """One tower layer of the GFL head on all levels (gfl_head.py:158-177,219-223): c = conv3x3(x) (no bias), y = ReLU(GroupNorm(c)).
    x_cat [N,A,Cin]; w [Cout,3,3,Cin] (OHWI).  Returns (c, y, mean_rstd).
    Where the convolution runs on the 128-couts-per-item three-limb Winograd kernel its output stage accumulates the group sums
    of its items and a tiny launch folds them to (mean, 1 / std) (erd_wino_conv3x3_x3_gn); the normalisation is the apply pass alone
    (erd_gn_relu_apply) -- the statistics pass over c and its zero fill disappear; everywhere else: conv_forward + gn_relu_forward."""
    _require_gpu(x_cat, w)
    N, A, Cin = x_cat.shape
    Cout = w.shape[0]
    c = torch.empty((N, A, Cout), dtype=x_cat.dtype, device=x_cat.device)
    xs, outs = level_views(x_cat, sizes), level_views(c, sizes)
    fused = (GN_FUSED and Cout == 256 and G == 32 and x_cat.dtype == torch.float32 and wino_ok(Cin, 3, 1, 1) and wino_x3()
             and (WINO_TRAIN_FWD if RECORDED else WINO_NOGRAD_FWD))
    ws_bytes = 0
    if fused:
        from ._lib import ConvSeg
        segs = (ConvSeg * len(xs))()
        for i, (x, o) in enumerate(zip(xs, outs)):
            _fill_seg(segs[i], x, o, x.shape[1], x.shape[2], None, None, None)
        ws_bytes = int(_lib.load().erd_wino_x3_gn_ws_bytes(segs, len(xs), Cout))      # 0: this launch runs on the 64-couts-per-item kernel
    if ws_bytes == 0:
        conv_forward(xs, w, outs, 3, 1, 1)
        y, mr = gn_relu_forward(c, gamma, beta, sizes, G, eps)
        return c, y, mr
    U = _wino_weights_cached(w)
    lv = make_levels(sizes)
    part = workspace("gn_part", ws_bytes, c.device)
    mr = torch.empty((N, lv.nseg, G, 2), dtype=torch.float32, device=c.device)
    flop = 2.0 * N * A * Cout * 9 * Cin
    nbytes = (4.0 * (x_cat.numel() + c.numel()) + U.numel() * U.element_size()) if _TIMING is not None else 0.0
    _timed_call("conv_wino_fwd_p", flop, "erd_wino_conv3x3_x3_gn", segs, len(xs), _p(U), Cin, Cout, None, None, 0,
                _p(_wino_sched(U.device)), _p(part), ws_bytes, _stream(), nbytes=nbytes,
                tag=f"px{N * A} {Cin}->{Cout} k3s1" if TIMING_DETAIL else "")
    call("erd_wino_gn_finalize", segs, len(xs), Cout, _p(part), _p(mr), eps, _stream())      # (its own call: not inside the conv's timed window)
    y = torch.empty_like(c)
    call("erd_gn_relu_apply", _p(c), _p(y), _p(gamma), _p(beta), _p(mr), N, A, Cout, G, C.byref(lv), _mt(c), _stream())
    return c, y, mr


def gn_relu_backward(c: Tensor, dy: Tensor, gamma: Tensor, beta: Tensor, mr: Tensor, sizes, G: int = 32,
                     dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None):
    """dgamma / dbeta: zero-initialised accumulators to add into (the parameters' flat gradient slots), else fresh ones"""
    assert c.is_contiguous() and dy.is_contiguous()
    N, A, Cc = c.shape
    lv = make_levels(sizes)
    stats = workspace("gn_stats", N * lv.nseg * G * 16, c.device)
    dc = torch.empty_like(c)
    dgamma = zeros_f32(gamma.numel(), c.device) if dgamma is None else dgamma
    dbeta = zeros_f32(beta.numel(), c.device) if dbeta is None else dbeta
    call("erd_gn_relu_bwd", _p(c), _p(dy), _p(gamma), _p(beta), _p(mr), _p(stats), _p(dc), _p(dgamma), _p(dbeta), N, A,
         Cc, G, C.byref(lv), _mt(c, dy), _stream())
    return dc, dgamma, dbeta


def upsample_add_(fine: Tensor, coarse: Tensor) -> None:
    _check_map(fine)
    _check_map(coarse)
    N, H, W, Cc = fine.shape
    call("erd_upsample2x_add", _p(fine), _p(coarse), N, H, W, Cc, coarse.shape[1], coarse.shape[2], fine.stride(0),
         coarse.stride(0), _mt(fine, coarse), _stream())


def upsample_add_bwd_(dfine: Tensor, dcoarse: Tensor) -> None:
    """dcoarse += adjoint(nearest-upsample)(dfine)"""
    _check_map(dfine)
    _check_map(dcoarse)
    N, H, W, Cc = dfine.shape
    call("erd_upsample2x_add_bwd", _p(dfine), _p(dcoarse), N, H, W, Cc, dcoarse.shape[1], dcoarse.shape[2],
         dfine.stride(0), dcoarse.stride(0), _mt(dfine, dcoarse), _stream())


def level_scale(x: Tensor, alphas: Tensor, sizes) -> Tensor:
    assert x.is_contiguous() and alphas.is_contiguous()
    N, A, Cc = x.shape
    lv = make_levels(sizes)
    y = torch.empty_like(x)
    call("erd_level_scale", _p(x), _p(alphas), _p(y), N, A, Cc, C.byref(lv), _stream())
    return y


def level_scale_bwd(x: Tensor, dy: Tensor, alphas: Tensor, sizes):
    assert x.is_contiguous() and dy.is_contiguous()
    N, A, Cc = x.shape
    lv = make_levels(sizes)
    dx = torch.empty_like(x)
    dal = torch.empty_like(alphas)
    call("erd_level_scale_bwd", _p(x), _p(dy), _p(alphas), _p(dx), _p(dal), N, A, Cc, C.byref(lv), _stream())
    return dx, dal


def sgd_momentum_(p: Tensor, g: Tensor, buf: Tensor, lr: float, momentum: float, weight_decay: float,
                  grad_scale: float, first_step: bool) -> None:
    assert p.is_contiguous() and g.is_contiguous() and buf.is_contiguous() and p.numel() % 4 == 0
    call("erd_sgd_momentum", _p(p), _p(g), _p(buf), p.numel(), lr, momentum, weight_decay, grad_scale,
         1 if first_step else 0, _stream())


# ---------------------------------------------------------------------------------------------
# ERS / anchors / ATSS / losses
# ---------------------------------------------------------------------------------------------
def ers_select(t_cls: Tensor, t_bbox: Tensor):
    """[N,A,Ccls], [N,A,68] -> dict(mask_cls, mask_bbox [N,A] u8; idx_cls, idx_bbox [N,A] i64 (first counts valid);
    counts [N,2] i32; thr [N,2] f32).  No host sync."""
    _require_gpu(t_cls, t_bbox)
    assert t_cls.is_contiguous() and t_bbox.is_contiguous()
    N, A, Cc = t_cls.shape
    dev = t_cls.device
    out = dict(mask_cls=torch.empty((N, A), dtype=torch.uint8, device=dev),
               mask_bbox=torch.empty((N, A), dtype=torch.uint8, device=dev),
               idx_cls=torch.empty((N, A), dtype=torch.int64, device=dev),
               idx_bbox=torch.empty((N, A), dtype=torch.int64, device=dev),
               counts=torch.empty((N, 2), dtype=torch.int32, device=dev),
               thr=torch.empty((N, 2), dtype=torch.float32, device=dev))
    ws = workspace("ers", N * 32 + 2 * N * A * 4, dev)
    call("erd_ers_select", _p(t_cls), _p(t_bbox), N, A, Cc, t_bbox.shape[2], _p(out["mask_cls"]), _p(out["mask_bbox"]),
         _p(out["idx_cls"]), _p(out["idx_bbox"]), _p(out["counts"]), _p(out["thr"]), _p(ws), _stream())
    return out


def _iarr(vals, ctype=C.c_int):
    return (ctype * len(vals))(*[int(v) for v in vals])


def grid_anchors(sizes, strides, device, octave_scale: int = 8) -> Tensor:
    A = sum(h * w for h, w in sizes)
    anchors = torch.empty((A, 4), dtype=torch.float32, device=device)
    call("erd_grid_anchors", _p(anchors), _iarr([h for h, _ in sizes]), _iarr([w for _, w in sizes]), _iarr(strides),
         len(sizes), octave_scale, _stream())
    return anchors


def _lvl_off(sizes):
    off = [0]
    for h, w in sizes:
        off.append(off[-1] + h * w)
    return _iarr(off, C.c_int64)


def atss_assign(anchors: Tensor, valid: Optional[Tensor], sizes, gt_boxes: Tensor, gt_labels: Tensor, gt_off: Tensor,
                N: int, max_gt: int, num_classes: int, topk: int = 9):
    A = anchors.shape[0]
    dev = anchors.device
    labels = torch.empty((N, A), dtype=torch.int64, device=dev)
    lw = torch.empty((N, A), dtype=torch.float32, device=dev)
    bt = torch.empty((N, A, 4), dtype=torch.float32, device=dev)
    npos = torch.empty((N,), dtype=torch.int32, device=dev)
    ws = workspace("atss", N * A * 8, dev)
    call("erd_atss_assign", _p(anchors), _p(valid), _lvl_off(sizes), len(sizes), A, _p(gt_boxes), _p(gt_labels),
         _p(gt_off), N, max_gt, topk, num_classes, _p(labels), _p(lw), _p(bt), _p(npos), _p(ws), _stream())
    return labels, lw, bt, npos


def gfl_losses_fwd(cls, bbox, anchors, labels, lw, bt, sizes, strides, c_old, c_all):
    N, A, _ = cls.shape
    dev = cls.device
    score = torch.empty((N, A), dtype=torch.float32, device=dev)
    wt = torch.empty((N, A), dtype=torch.float32, device=dev)
    sums = torch.empty((len(sizes), 4), dtype=torch.float64, device=dev)
    call("erd_gfl_losses_fwd", _p(cls), _p(bbox), _p(anchors), _p(labels), _p(lw), _p(bt), _lvl_off(sizes),
         _iarr(strides), len(sizes), N, A, c_old, c_all, _p(score), _p(wt), _p(sums), _stream())
    return score, wt, sums


def gfl_losses_bwd(cls, bbox, anchors, labels, lw, bt, sizes, strides, c_old, c_all, score, wt, coef):
    N, A, _ = cls.shape
    dcls = torch.empty_like(cls)
    dbbox = torch.empty_like(bbox)
    call("erd_gfl_losses_bwd", _p(cls), _p(bbox), _p(anchors), _p(labels), _p(lw), _p(bt), _lvl_off(sizes),
         _iarr(strides), len(sizes), N, A, c_old, c_all, _p(score), _p(wt), _p(coef), _p(dcls), _p(dbbox), _stream())
    return dcls, dbbox


def l2_distill(s_cls, t_cls, idx_cls, counts, c_old):
    N, A, c_s = s_cls.shape
    sums = torch.empty((N,), dtype=torch.float64, device=s_cls.device)
    call("erd_l2_distill", _p(s_cls), _p(t_cls), _p(idx_cls), _p(counts), N, A, c_s, t_cls.shape[2], c_old, _p(sums),
         _stream())
    return sums


def l2_distill_bwd_(s_cls, t_cls, idx_cls, counts, coef, c_old, dcls):
    N, A, c_s = s_cls.shape
    call("erd_l2_distill_bwd", _p(s_cls), _p(t_cls), _p(idx_cls), _p(counts), _p(coef), N, A, c_s, t_cls.shape[2],
         c_old, _p(dcls), _stream())


def distill_nms(t_cls, t_bbox, anchors, idx_bbox, counts, iou_thr: float = 0.005):
    N, A, c_t = t_cls.shape
    dev = t_cls.device
    keep = torch.empty((N, A), dtype=torch.uint8, device=dev)
    kcnt = torch.empty((N,), dtype=torch.int32, device=dev)
    nbytes = N * A * 32
    ws = workspace("nms", nbytes, dev)
    call("erd_distill_nms", _p(t_cls), _p(t_bbox), _p(anchors), _p(idx_bbox), _p(counts), N, A, c_t, iou_thr, _p(keep),
         _p(kcnt), _p(ws), C.c_size_t(nbytes), _stream())
    return keep, kcnt


def predict_topk(cls: Tensor, bbox: Tensor, anchors: Tensor, sizes, strides, img_hw: Tensor, score_thr: float,
                 nms_pre: int):
    """per (image, level) top-nms_pre candidates above score_thr, decoded (gfl_head.py:408-502).
    Returns boxes [N, L*nms_pre, 4], scores, labels (int32) [N, L*nms_pre], num [N] (int32), level-major."""
    N, A, Cc = cls.shape
    dev = cls.device
    L = len(sizes)
    cols = L * nms_pre
    boxes = torch.empty((N, cols, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, cols), dtype=torch.float32, device=dev)
    labels = torch.empty((N, cols), dtype=torch.int32, device=dev)
    num = torch.empty((N,), dtype=torch.int32, device=dev)
    nbytes = int(_lib.load().erd_predict_ws_bytes(N, L, nms_pre))
    ws = workspace("predict", nbytes, dev)
    lv = make_levels(sizes)
    call("erd_predict_topk", _p(cls), _p(bbox), _p(anchors), N, A, Cc, C.byref(lv), _iarr(strides), _p(img_hw),
         float(score_thr), int(nms_pre), _p(boxes), _p(scores), _p(labels), _p(num), _p(ws), C.c_size_t(nbytes),
         _stream())
    return boxes, scores, labels, num


def predict_nms(boxes: Tensor, scores: Tensor, labels: Tensor, num: Tensor, inv_scale: Tensor, min_bbox_size: float,
                iou_thr: float, max_per_img: int):
    """rescale + size filter + class-offset NMS + top max_per_img (base_dense_head.py:424-486).
    Returns dets [N, max_per_img, 5], det_labels [N, max_per_img] (int64), det_num [N] (int32)."""
    N, cols, _ = boxes.shape
    dev = boxes.device
    dets = torch.zeros((N, max_per_img, 5), dtype=torch.float32, device=dev)
    det_labels = torch.zeros((N, max_per_img), dtype=torch.int64, device=dev)
    det_num = torch.empty((N,), dtype=torch.int32, device=dev)
    nbytes = N * cols * 48
    ws = workspace("predict_nms", nbytes, dev)
    call("erd_predict_nms", _p(boxes), _p(scores), _p(labels), _p(num), N, cols, _p(inv_scale), float(min_bbox_size),
         float(iou_thr), int(max_per_img), _p(dets), _p(det_labels), _p(det_num), _p(ws), C.c_size_t(nbytes), _stream())
    return dets, det_labels, det_num


def kd_kl(s_bbox, t_bbox, s_cls, keep, c_old, T):
    N, A, c_s = s_cls.shape
    sums = torch.empty((N,), dtype=torch.float64, device=s_cls.device)
    call("erd_kd_kl", _p(s_bbox), _p(t_bbox), _p(s_cls), _p(keep), N, A, c_s, c_old, float(T), _p(sums), _stream())
    return sums


def kd_kl_bwd_(s_bbox, t_bbox, s_cls, keep, coef, c_old, T, dbbox):
    N, A, c_s = s_cls.shape
    call("erd_kd_kl_bwd", _p(s_bbox), _p(t_bbox), _p(s_cls), _p(keep), _p(coef), N, A, c_s, c_old, float(T), _p(dbbox),
         _stream())


def loss_finalize(lvl_sums, avg, l2_sums, kd_sums, counts, nlvl, N, c_old, w_dist, lw_cls, lw_bbox, lw_dfl, lw_ld,
                  upstream: Optional[Tensor], want_losses: bool, want_coef: bool):
    dev = lvl_sums.device
    losses = torch.empty((3 * nlvl + 2 * N,), dtype=torch.float32, device=dev) if want_losses else None
    coef = torch.empty((4 * nlvl + 2 * N,), dtype=torch.float32, device=dev) if want_coef else None
    call("erd_loss_finalize", _p(lvl_sums), _p(avg), _p(l2_sums), _p(kd_sums), _p(counts), nlvl, N, c_old,
         float(w_dist), float(lw_cls), float(lw_bbox), float(lw_dfl), float(lw_ld), _p(upstream), _p(losses), _p(coef),
         _stream())
    return losses, coef


def loss_avg(num_pos: Tensor, lvl_sums: Tensor) -> Tensor:
    avg = torch.empty((2,), dtype=torch.float32, device=lvl_sums.device)
    call("erd_loss_avg", _p(num_pos), num_pos.numel(), _p(lvl_sums), lvl_sums.shape[0], _p(avg), _stream())
    return avg
