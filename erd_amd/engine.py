"""Training-step engine for the ERD path: what mmengine's ``model.train_step`` / ``OptimWrapper`` /
``MMDistributedDataParallel`` do around ``model(inputs, data_samples, mode='loss')``,
re-designed for one process per MI355X:

  * trainable parameters live in ONE flat fp32 buffer (and their gradients in another) so that the
    SGD(momentum, weight-decay) update is a single HIP launch and the data-parallel gradient mean is a
    handful of large RCCL all-reduces on contiguous bucket slices -- no flatten/unflatten copies;
  * buckets are laid out in *backward* order (head -> neck -> layer4 -> layer2) and each bucket's
    all-reduce is issued from a post-accumulate hook as soon as its last gradient lands, i.e. it rides
    over xGMI while the rest of the backward is still computing;
  * the optimizer step of iteration t is deferred to the start of iteration t+1, behind the (frozen)
    teacher's forward of batch t+1 which runs on a side HIP stream -- the teacher never depends on the
    update, so the tail of the all-reduce and the SGD launch hide under it (north_star).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import os
import time
import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from .dist_utils import all_reduce_sum_
from .modules import GFLIncrementERD, parse_losses
from .structures import unpack_gt_instances

Tensor = torch.Tensor
ALIGN = 64  # floats: every parameter starts on a 256-B boundary (kernels read weights as float4)


def _storage_view(flat: Tensor, off: int, p: Tensor) -> Tensor:
    """a view of flat[off:off+numel] with p's logical shape AND p's physical (dense) layout."""
    n = p.numel()
    if p.dim() == 4:
        O, I, kh, kw = p.shape
        v = p.detach().permute(0, 2, 3, 1)
        if v.is_contiguous():      # channels_last parameter: keep [O][kh][kw][I]
            return flat[off:off + n].view(O, kh, kw, I).permute(0, 3, 1, 2)
    return flat[off:off + n].view(p.shape)


class FlatParams:
    def __init__(self, named_params: Sequence, device, bucket_bytes: int = 32 << 20, tail_bytes: int = 4 << 20, group_key=None):
        """`group_key(name)`: parameters with the same key are never split between two buckets (ERDTrainer: one key per
        ResNet block / per neck / per head -- the unit whose backward launches are issued together, see BucketedGradSync)."""
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.offsets, self.total = offs, total
        self.data = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.momentum = torch.zeros(total, dtype=torch.float32, device=device)
        for p, off in zip(self.params, offs):
            v = _storage_view(self.data, off, p)
            v.copy_(p.detach())
            p.data = v
            p.grad = _storage_view(self.grad, off, p)
            p._erd_sink = True        # backward kernels may write this slot directly (functional._sink)
        self.data_bf16: Optional[Tensor] = None      # bf16 shadow of `data` (bf16 matrix-core mode), see refresh_shadow
        # units: runs of parameters that stay together (one parameter each without a group key), in layout order (= backward order)
        ends = offs[1:] + [total]
        units: List[list] = []
        prev = object()
        for i, name in enumerate(self.names):
            key = group_key(name) if group_key is not None else i
            if units and key == prev:
                units[-1].append(i)
            else:
                units.append([i])
            prev = key
        # buckets over the flat range: whole units until `bucket_bytes` are reached
        self.buckets: List[tuple] = []
        cur: List[list] = []
        for u in units:
            cur.append(u)
            if (ends[u[-1]] - offs[cur[0][0]]) * 4 >= bucket_bytes or u is units[-1]:
                self.buckets.append(cur)
                cur = []
        # the LAST bucket's all-reduce and update are the ones nothing of the backward pass is left to hide: with the teacher
        # of step t+1 running next to backward(t) they sit between backward(t) and forward(t+1).  Keep them short: the tail
        # of the layout (the parameters whose gradients land last: the first trainable block(s)) becomes its own bucket of
        # at most `tail_bytes` (GFL-R50: layer2's first three blocks, 3.6 MB, instead of 19 MB)
        size = lambda us: (ends[us[-1][-1]] - offs[us[0][0]]) * 4
        if self.buckets and len(self.buckets[-1]) > 1:
            us = self.buckets[-1]
            cut = len(us)
            while cut > 1 and size(us[cut - 1:]) <= tail_bytes:
                cut -= 1
            if cut < len(us) and size(us[cut:]) <= tail_bytes < size(us):
                self.buckets[-1:] = [us[:cut], us[cut:]]
        self.buckets = [(offs[us[0][0]], ends[us[-1][-1]], [i for u in us for i in u]) for us in self.buckets]
        self.bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b

    def zero_grad(self) -> None:
        self.grad.zero_()

    def refresh_shadow(self, bucket: Optional[int] = None) -> None:
        """bf16 matrix-core mode: ONE conversion launch over the flat parameter buffer (or one bucket of it) after each update
        instead of one per convolution; every 4-d parameter gets a bf16 twin view (same shape and strides) that the conv
        wrappers pick up as long as the parameter's version counter has not moved since."""
        if K.COMPUTE != "bf16":
            return
        if self.data_bf16 is None:
            self.data_bf16 = torch.empty(self.total, dtype=torch.bfloat16, device=self.data.device)
            if bucket is not None:       # (first use from a per-bucket update: the other buckets follow within the same step)
                K.to_bf16_into(self.data, self.data_bf16)
        if bucket is None:
            K.to_bf16_into(self.data, self.data_bf16)
            members = range(len(self.params))
        else:
            s, e, members = self.buckets[bucket]
            K.to_bf16_into(self.data[s:e], self.data_bf16[s:e])
        for i in members:
            p = self.params[i]
            if p.dim() == 4:
                p._erd_shadow = (_storage_view(self.data_bf16, self.offsets[i], p), p._version)


class BucketedGradSync:
    """Data-parallel gradient SUM over ranks on the flat gradient buffer: one async all-reduce per bucket, issued
    from a post-accumulate-grad hook when the bucket's last gradient has landed (so it overlaps the rest of the
    backward).  Device-agnostic (RCCL on GPUs, gloo in the CPU tests); the 1/world scaling is folded into the SGD
    kernel's `grad_scale`.

    With `on_bucket` the sync also drives a per-bucket UPDATE: `on_bucket(b)` runs once bucket b's summed gradient is final
    (ERDTrainer: SGD on the bucket's slice, its BN folds, its prepared weight buffers), so that at the step boundary only the
    small tail bucket is left.  The update rewrites what the bucket's own backward launches still read (weights, folded BN
    scales, transposed weights), and a parameter reports as soon as ITS gradient kernel is queued -- the input-gradient launch
    of the same convolution may follow.  So a complete bucket is released by the first report that comes from ANOTHER bucket
    (buckets hold whole blocks -- FlatParams(group_key=...) -- and a block's launches are issued together, hence all queued by
    then), or by wait().  With `stream` (a side HIP stream) collective and update are queued there, behind everything the
    producing streams hold at that moment; the producing streams are never made to wait.  `reduce=False`: a single rank --
    the hooks only drive the update."""

    def __init__(self, flat: FlatParams, streams: Sequence = (), reduce: bool = True, on_bucket=None, stream=None, live_streams=None):
        self.flat = flat
        self.streams = list(streams)     # HIP streams that may hold gradient-producing kernels of one backward pass
        # ... as known at construction.  A step may run under ANOTHER current stream (`with torch.cuda.stream(s)`, a capture's
        # warm-up stream, a caller-owned stream): `live_streams()` names the producers of the moment a bucket is released (the
        # current stream and the auxiliary streams keyed on it), and both sets are joined -- the update must never overtake a
        # gradient kernel, nor rewrite weights an input-gradient launch still reads
        self.live_streams = live_streams
        self.reduce = reduce
        self.on_bucket = on_bucket
        self.stream = stream
        self._works: List = []
        self._remaining: List[int] = []
        self.late_buckets = 0            # buckets that were not complete when backward ended (diagnostic)
        self.missing: List[str] = []     # ... and the parameters that had not reported a gradient by then (last occurrence)
        self._seen: List[bool] = []
        self.repeats = set()             # parameters that reported more than once in a step (diagnostic: the contract is once)
        self._warned_late = False
        self._next = 0
        self.issued_in_backward = 0      # buckets released by a hook (not by wait()) in the last step (diagnostic)
        for i, p in enumerate(flat.params):
            hook = self._make_hook(i)
            p.register_post_accumulate_grad_hook(hook)
            p._erd_sink_notify = hook      # gradients written straight into the flat slot bypass AccumulateGrad

    def arm(self) -> None:
        self._remaining = [len(m) for (_, _, m) in self.flat.buckets]
        self._seen = [False] * len(self.flat.params)
        self._works = []
        self._next = 0                   # collectives are issued strictly in bucket order: the same order on every rank
        self.issued_in_backward = 0

    def disarm(self) -> None:
        """the coming backward is not this sync's business (a captured whole-step graph): hooks do nothing until arm()"""
        self._remaining = []

    def _producers(self) -> List:
        dev = self.flat.grad.device
        if dev.type != "cuda":
            return []
        sts, seen = [], set()
        for st in list(self.streams) + [torch.cuda.current_stream(dev)] + (list(self.live_streams()) if self.live_streams else []):
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                sts.append(st)
        return sts

    def _issue(self, b: int) -> None:
        s, e, _ = self.flat.buckets[b]
        g = self.flat.grad[s:e]
        if self.stream is not None:
            # behind every kernel the producing streams hold now (a bucket's gradients come from kernels on several streams:
            # the two head towers run their backward on two, the backbone's weight gradients trail on a third)
            for st in self._producers():
                if st.cuda_stream != self.stream.cuda_stream:
                    self.stream.wait_stream(st)
            with torch.cuda.stream(self.stream):
                if self.reduce:
                    w = all_reduce_sum_(g, async_op=True)
                    if w is not None:
                        w.wait()         # (RCCL: orders this stream behind the collective; the host does not block)
                if self.on_bucket is not None:
                    self.on_bucket(b)
            return
        if self.streams:
            cs = torch.cuda.current_stream(self.flat.grad.device)
            for st in self._producers():
                if st.cuda_stream != cs.cuda_stream:
                    cs.wait_stream(st)
        w = all_reduce_sum_(g, async_op=True) if self.reduce else None
        if self.on_bucket is not None:
            if w is not None:
                w.wait()
            self.on_bucket(b)
        elif w is not None:
            self._works.append(w)

    def _make_hook(self, i: int):
        def hook(_p):
            if not self._remaining:
                return
            if self._seen[i]:            # (one notification per parameter and step: a second one must not drive the countdown
                self.repeats.add(self.flat.names[i])      # below zero -- the in-order issue loop waits for exactly zero)
                return
            self._seen[i] = True
            b = self.flat.bucket_of[i]
            self._remaining[b] -= 1
            # issue every complete bucket at the head of the queue.  A bucket that completes before an earlier one (a
            # parameter without a gradient on THIS rank only) waits for it, so that all ranks enqueue the collectives of
            # one communicator in the same order whatever their local completion order is.  With a per-bucket update the
            # bucket this report belongs to stays queued (see the class docstring)
            hold = b if self.on_bucket is not None else -1
            while self._next < len(self._remaining) and self._remaining[self._next] == 0 and self._next != hold:
                self._issue(self._next)
                self._next += 1
                self.issued_in_backward += 1
        return hook

    def issue_rest(self) -> None:
        """backward has ended (every producing launch is queued): release what is left, in bucket order"""
        # a bucket whose countdown never reached zero (a parameter without a gradient this step, or a broken
        # "one notification per parameter" contract) would leave rank-local gradients in the flat buffer and let the
        # ranks diverge silently: it is issued here all the same
        late = sum(1 for r in self._remaining[self._next:] if r != 0) if self._remaining else 0
        while self._remaining and self._next < len(self._remaining):
            self._issue(self._next)
            self._next += 1
        if late:
            self.late_buckets += late
            self.missing = [self.flat.names[i] for i, seen in enumerate(self._seen) if not seen]
            if not self._warned_late:
                self._warned_late = True
                import warnings
                warnings.warn(f"BucketedGradSync: {late} of {len(self._remaining)} gradient buckets were not complete when backward "
                              "ended (a parameter without a gradient this step?); their all-reduce was issued late, in bucket order.  "
                              f"Parameters that did not report a gradient: {self.missing[:8]}{' ...' if len(self.missing) > 8 else ''}")
        self._remaining = []

    def wait(self) -> None:
        self.issue_rest()
        for w in self._works:
            w.wait()
        self._works = []
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.grad.device).wait_stream(self.stream)


class TeacherGraphs:
    """hipGraph capture of the frozen teacher's half of the step (shared frozen trunk, forward, ERS, NMS of the selected
    boxes): the teacher never changes, so its ~150 launches per batch are recorded once per input shape and replayed as one
    graph launch (BASELINE.json configs[4]: mixed-resolution batches -> one graph per padded shape, built on first use).
    Outputs live in the graph's static buffers and are overwritten by the next replay of the SAME graph.  Each shape
    therefore has TWO graphs (`slot` = parity of the step): step t's backward pass still reads slot t % 2 (teacher logits,
    ERS lists, the trunk map the student's first trainable block saved) while the look-ahead replay for step t+1 fills the
    other one; slot t % 2 is replayed again for step t+2, which the trainer orders behind step t+1's losses and thus
    behind everything of step t.
    The shared frozen trunk (GFLIncrementERD.shares_trunk) is part of the capture: the student reads it from the static
    buffer, ordered by an event recorded after the replay (an event cannot be recorded for another stream inside a
    capture)."""

    def __init__(self, model: GFLIncrementERD, max_graphs: int = 32):
        self.model, self.max_graphs = model, max_graphs
        self.graphs: Dict[tuple, tuple] = {}

    def run(self, inputs: Tensor, slot: int = 0):
        """call with the stream the teacher should run on as the current stream"""
        key = (tuple(inputs.shape), inputs.device.index, K.COMPUTE, slot & 1)
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            stream = torch.cuda.current_stream(inputs.device)
            K.pin_workspaces()               # graphs hold workspace addresses: outgrown buffers are retired, not freed
            static_in = inputs.clone()
            with torch.no_grad():
                self.model.teacher_pass(static_in, share_trunk="static")      # eager warm-up: workspaces, anchor / folded-BN caches
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream), torch.no_grad():
                out = self.model.teacher_pass(static_in, share_trunk="static")
            ent = self.graphs[key] = (graph, static_in, out)
        graph, static_in, out = ent
        static_in.copy_(inputs, non_blocking=True)
        graph.replay()
        out.targets = None
        if out.trunk is not None:            # the hand-over event of the shared trunk: after the replay, on the replaying stream
            out.trunk_event = torch.cuda.Event()
            out.trunk_event.record()
        return out


class ERDTrainer:
    """One optimisation step of the ERD incremental detector per call (teacher fwd -> ERS -> student fwd ->
    losses -> backward -> gradient mean over ranks -> SGD)."""

    def __init__(self, model: nn.Module, lr: float = 0.01, momentum: float = 0.9, weight_decay: float = 1e-4,
                 base_batch_size: int = 16, batch_size_per_gpu: Optional[int] = None, auto_scale_lr: bool = True,
                 warmup_iters: int = 500, warmup_start_factor: float = 0.001, bucket_mb: int = 32,
                 overlap_teacher: bool = True, teacher_graph: bool = False, step_graph: bool = False):
        self.model = model
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.distributed else 1
        self.cu_reserve = 0                  # CUs the grids leave to RCCL's resident kernels (tune_cu_reserve; 0 at one rank)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("ERDTrainer needs the model on the GPU (erd_amd has no CPU path)")
        self.device = dev
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not n.startswith("ori_model.")]
        named.reverse()                      # backward order: head first, layer2 last
        self.flat = FlatParams(named, dev, bucket_mb << 20, group_key=self._block_of)
        self.flat.refresh_shadow()
        self.base_lr = lr * (self.world * batch_size_per_gpu / base_batch_size
                             if (auto_scale_lr and batch_size_per_gpu) else 1.0)   # auto_scale_lr (config :116)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.warmup_iters, self.warmup_start = warmup_iters, warmup_start_factor
        self.iter = 0
        self.epoch_factor = 1.0              # MultiStepLR factor of the current epoch (set by the runner)
        self.lr_factor = None                # optional callable iter -> factor replacing the built-in warm-up
        self.last_lr = self.base_lr
        self._pending_lr = self.base_lr
        self._first = True
        self._pending = False                # an un-applied gradient sits in flat.grad
        self._teacher_ahead = None           # (inputs, TeacherOut) of the following step (train_step(next_batch=...))
        self.prefold = Fn.BnPrefold(model) if os.environ.get("ERD_BN_PREFOLD", "1") != "0" else None
        self.prep = K.ParamPrep(self.device) if os.environ.get("ERD_PARAM_PREP", "1") != "0" and self.prefold is not None else None
        for p in self.flat.params:           # the wrappers find the prepared buffers through the parameter they belong to
            p._erd_prep = self.prep          # (kernels._prep_of): several trainers in one process do not see each other's
        if self.prep is not None:
            self.prep.on_stale = self._drop_step_graphs
        self.is_erd = isinstance(model, GFLIncrementERD)
        self.overlap_teacher = overlap_teacher and self.is_erd
        # the update per gradient bucket, next to the rest of the backward pass (BucketedGradSync): SGD on the bucket's slice
        # of the flat buffers, then its BN folds and its prepared weight buffers, on a side stream; what is left at the step
        # boundary is the tail bucket (<= 4 MB).  At one rank too: the ~0.5 ms of update launches leave the critical path
        # Default: on in the fp32 modes (level at one rank: 113.35 against 113.23 img/s, three alternating pairs), OFF in the bf16 mode:
        # there the step is close to host-bound (14.6 ms of issue work under a 19.1 ms step, profiles/r06_bf16_host_breakdown.txt) and
        # the per-bucket launches issued from the autograd thread's hooks delay the backward pass's own launches -- 208.7 img/s with,
        # 216.4 without, three alternating pairs on one box (profiles/r06_bf16_bucket_update_ab.txt; this is the "unexplained" 215 -> 208
        # of round 5).  The bucket ALL-REDUCES stay asynchronous either way; ERD_BUCKET_UPDATE=0 / 1 overrides.
        bu_default = "0" if K.COMPUTE == "bf16" else "1"
        self.bucket_update = (os.environ.get("ERD_BUCKET_UPDATE", bu_default) != "0" and not step_graph and self.prefold is not None
                              and self.prep is not None)
        self.sync = None
        if self.distributed or self.bucket_update:
            producers = [torch.cuda.current_stream(dev), Fn.aux_stream(dev), Fn.trail_stream(dev)]
            live = lambda: [Fn.aux_stream(dev), Fn.trail_stream(dev)]      # (keyed on the stream that is current when a bucket is released)
            if self.bucket_update:
                bucket_of = {id(p): self.flat.bucket_of[i] for i, p in enumerate(self.flat.params)}
                self.prefold.set_groups(lambda p: bucket_of.get(id(p)))
                self.prep.group_of = lambda p: bucket_of.get(id(p))
                self.sync = BucketedGradSync(self.flat, streams=producers, reduce=self.distributed, on_bucket=self._update_bucket,
                                             stream=torch.cuda.Stream(device=dev), live_streams=live)
            else:
                self.sync = BucketedGradSync(self.flat, streams=producers, live_streams=live)
        self.side = torch.cuda.Stream(device=dev) if self.overlap_teacher else None
        # whole-step hipGraph (one per input shape): everything between two SGD updates -- teacher, ERS, NMS, targets,
        # student forward, losses, backward on all streams -- is recorded once and replayed as ONE launch; the step is
        # free of host synchronisation, so nothing in it depends on the host.  ~800 launches of Python / ctypes issue work
        # per step disappear (the bf16 matrix-core mode is bound by exactly that: 30 ms of issue time for a 30 ms step).
        # Single-GPU ERD steps only (a captured RCCL all-reduce could not be tested on this pool).
        self.step_graph = bool(step_graph) and self.overlap_teacher and not self.distributed
        self._step_graphs: Dict[tuple, tuple] = {}
        if teacher_graph and not self.overlap_teacher:
            raise ValueError("teacher_graph needs the ERD detector with overlap_teacher=True")
        self.teacher_graphs = TeacherGraphs(model) if teacher_graph else None

    def _drop_step_graphs(self) -> None:
        """captured steps read the prepared weight buffers by address: once those stop vouching for the parameters
        (re-homed storage, a parameter edited behind the trainer's back) the graphs are recorded anew on next use"""
        self._step_graphs.clear()

    # -- schedule (schedule_1x.py:7-17: LinearLR warm-up; MultiStep handled by the caller per epoch) ------------
    def lr_at(self, it: int, epoch_factor: float = 1.0) -> float:
        if self.lr_factor is not None:
            return self.base_lr * self.lr_factor(it) * epoch_factor
        f = 1.0
        if it < self.warmup_iters:
            f = self.warmup_start + (1.0 - self.warmup_start) * it / max(self.warmup_iters - 1, 1)
        return self.base_lr * f * epoch_factor

    @staticmethod
    def _block_of(name: str) -> str:
        """the unit of the backward pass a parameter belongs to (FlatParams(group_key=...)): a ResNet block (one
        functional._bottleneck_backward call: resnet.py:263-302), the neck, the head.  Buckets hold whole units"""
        parts = name.split(".")
        return ".".join(parts[:3]) if parts[0] == "backbone" and len(parts) > 3 else parts[0]

    def _update_bucket(self, b: int) -> None:
        """BucketedGradSync.on_bucket: the summed gradient of bucket b is final and every launch that reads the bucket's
        weights / folded scales / prepared buffers is queued ahead of the current (update) stream"""
        s, e, _ = self.flat.buckets[b]
        K.sgd_momentum_(self.flat.data[s:e], self.flat.grad[s:e], self.flat.momentum[s:e], self._pending_lr, self.momentum,
                        self.weight_decay, 1.0 / self.world, self._first)
        self.flat.refresh_shadow(b)
        self.prefold.run_group(b)
        if self.prefold.valid[0]:
            self.prep.run_group(b)
        else:
            self.prep.invalidate()

    def _apply_pending(self) -> None:
        """wait for the bucket all-reduces of the previous backward, then ONE fused SGD launch -- or, with the per-bucket
        update, release the buckets that are left (the tail) and join the update stream."""
        if not self._pending:
            return
        if self.bucket_update:
            self.sync.wait()
            self._first = False
            self._pending = False
            return
        if self.sync is not None:
            self.sync.wait()
        K.sgd_momentum_(self.flat.data, self.flat.grad, self.flat.momentum, self._pending_lr, self.momentum,
                        self.weight_decay, 1.0 / self.world, self._first)
        self._first = False
        self._pending = False
        self.flat.refresh_shadow()
        if self.prefold is not None:
            self.prefold.run()               # every trainable BN of the student folded for the coming step, one launch
        if self.prep is not None:            # transposed weights / Winograd weight images of the coming step, two launches
            if self.prefold is not None and self.prefold.valid[0]:
                self.prep.run()
            else:
                self.prep.invalidate()

    def flush(self) -> None:
        self._apply_pending()

    def tune_cu_reserve(self, batches, candidates: Sequence[int] = (0, 4, 8), steps: int = 3, timer=None) -> dict:
        """Data parallel only (a no-op report at one rank): the kernels' persistent / stream-K / one-round grids are whole-chip static
        grids, and RCCL's resident kernels take CUs away from them -- any resident foreign workgroup cost a whole dispatch round (-14 %,
        profiles/r05_cu_theft_step.txt).  With the process group live, run `steps` real training steps at every candidate reserve
        (kernels.set_cu_reserve: grids sized for CUs - reserve), time them, let all ranks agree on the candidate whose SLOWEST rank was
        fastest (dist_utils.agree_on_fastest) and keep it.  `batches`: a sequence of (inputs, data_samples) to cycle through;
        `timer(reserve) -> seconds` replaces the measurement (tests).  The steps are ordinary optimisation steps (warm-up iterations)."""
        if self.world == 1:
            return {"cu_reserve": 0, "probed": False, "note": "one rank: no resident collective kernels, the grids keep the whole chip"}
        from .dist_utils import agree_on_fastest
        secs = []
        for r in candidates:
            K.set_cu_reserve(int(r))
            if timer is not None:
                secs.append(float(timer(int(r))))
                continue
            self.train_step(*batches[0])         # descriptors / workspaces of this grid size
            self.flush()
            torch.cuda.synchronize(self.device)
            dist.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                self.train_step(*batches[i % len(batches)])
            self.flush()
            torch.cuda.synchronize(self.device)
            secs.append(time.perf_counter() - t0)
        best = agree_on_fastest(secs)
        K.set_cu_reserve(int(candidates[best]))
        self.cu_reserve = int(candidates[best])
        return {"cu_reserve": self.cu_reserve, "probed": True, "candidates": [int(c) for c in candidates],
                "local_seconds_per_step": [round(t / max(steps, 1), 5) for t in secs], "steps_per_candidate": steps}

    def join_streams(self) -> None:
        """order every stream the trainer launches on (teacher side stream, tower / trailing weight-gradient streams, the update
        stream) in front of the CURRENT stream: afterwards a wait on the current stream alone covers the whole step (bench.py
        --occupy-cus cannot use a device-wide wait, which would sit out the foreign spin kernel)"""
        cur = torch.cuda.current_stream(self.device)
        streams = [Fn.aux_stream(self.device), Fn.trail_stream(self.device), self.side]
        if self.sync is not None and getattr(self.sync, "stream", None) is not None:
            streams.append(self.sync.stream)
        for st in streams:
            if st is not None and st != cur:
                cur.wait_stream(st)

    # -- optimizer state in torch.optim.SGD's state_dict layout (what the reference's checkpoints hold) ---------
    def optimizer_state_dict(self) -> dict:
        self.flush()
        index = {id(p): i for i, p in enumerate(self.model.parameters())}
        state = {}
        if not self._first:
            for p, off in zip(self.flat.params, self.flat.offsets):
                buf = _storage_view(self.flat.momentum, off, p)
                state[index[id(p)]] = dict(momentum_buffer=buf.detach().cpu().contiguous().clone())
        group = dict(lr=self.last_lr, momentum=self.momentum, dampening=0, weight_decay=self.weight_decay, nesterov=False,
                     maximize=False, foreach=None, differentiable=False, initial_lr=self.base_lr,
                     params=list(range(len(index))))
        return dict(state=state, param_groups=[group])

    def load_optimizer_state_dict(self, sd: dict) -> None:
        self.flush()
        params = list(self.model.parameters())
        slot = {id(p): (p, off) for p, off in zip(self.flat.params, self.flat.offsets)}
        loaded = 0
        for i, st in sd.get("state", {}).items():
            p = params[int(i)]
            if id(p) not in slot or "momentum_buffer" not in st or st["momentum_buffer"] is None:
                continue
            p, off = slot[id(p)]
            _storage_view(self.flat.momentum, off, p).copy_(st["momentum_buffer"].to(self.device))
            loaded += 1
        self._first = loaded == 0

    # -- the step ------------------------------------------------------------------------------------------------------
    GT_CAPACITY = 64          # ground-truth boxes per image a captured step has room for (more: that batch runs eagerly)

    def _graph_body(self, st) -> Tensor:
        """one step minus the SGD update on the static buffers `st` (captured by _train_step_graph); returns the vector
        of logged scalars"""
        model, dev = self.model, self.device
        cur = torch.cuda.current_stream(dev)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side), torch.no_grad():
            t = model.teacher_pass(st.x)
            t.targets = model.bbox_head._targets_packed(t.sizes, st.gb, st.gl, st.goff, self.GT_CAPACITY, st.metas, dev)
        self.flat.zero_grad()
        K.zero_arena_begin(dev)
        if t.trunk is not None:
            cur.wait_event(t.trunk_event)
            t.trunk[0].record_stream(cur)
        with K.distillation_forward(K.WINO_FROZEN_TRUNK):
            s_cls, s_bbox, sizes = model._forward_cat(st.x, trunk=t.trunk)
        cur.wait_stream(self.side)
        for v in t.tensors():
            v.record_stream(cur)
        losses = model.bbox_head.loss_cat(t.t_cls, t.t_bbox, s_cls, s_bbox, sizes, None, t.ers, t.keep,
                                          model.ori_num_classes, model.dist_loss_weight, targets=t.targets)
        total, log_vars = parse_losses(losses)
        total.backward()
        Fn.trail_join(dev)
        K.zero_arena_end()
        st.names = list(log_vars.keys())
        return torch.stack([v.reshape(()) for v in log_vars.values()])

    def _train_step_graph(self, inputs: Tensor, data_samples):
        """-> log_vars, or None when this batch does not fit the captured form (too many boxes): the caller runs it eagerly"""
        from types import SimpleNamespace
        dev = self.device
        gts, _, metas = unpack_gt_instances(data_samples)
        N = inputs.shape[0]
        counts = [int(g.bboxes.shape[0]) for g in gts]
        if max(counts + [0]) > self.GT_CAPACITY:
            return None
        key = (tuple(inputs.shape), tuple(tuple(m["pad_shape"][:2]) for m in metas), inputs.device.index, K.COMPUTE)
        ent = self._step_graphs.get(key)
        cur = torch.cuda.current_stream(dev)
        if ent is None:
            st = SimpleNamespace(x=torch.empty_like(inputs), gb=torch.zeros((N * self.GT_CAPACITY, 4), device=dev),
                                 gl=torch.zeros((N * self.GT_CAPACITY,), dtype=torch.long, device=dev),
                                 goff=torch.zeros((N + 1,), dtype=torch.int32, device=dev),
                                 metas=[dict(pad_shape=tuple(m["pad_shape"])) for m in metas], names=None)
            self._fill_static(st, inputs, gts, counts)
            K.pin_workspaces()               # graphs hold workspace addresses: outgrown buffers are retired, not freed
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(cur)
            Fn.CAPTURE_ORIGIN = s.cuda_stream        # joins only into the origin stream: see functional.CAPTURE_ORIGIN
            try:
                with torch.cuda.stream(s):   # eager warm-up on the capture stream: workspaces, caches, auxiliary streams
                    for _ in range(2):
                        self._graph_body(st)
                    # The warm-up has REGISTERED the step's derived-weight recipes (folded BN scales, transposed / limb-split weights,
                    # Winograd weight images) but they only vouch for the parameters once the batched preparation has run -- until round
                    # 6 the capture therefore recorded the ~250 per-use preparation launches of a step (42 bn_fold, 45 wino_weight_x3,
                    # 58 weight_transpose, 73 split3, ...: `profiles/r06_graph_timelines.txt`) and replayed them every step, which is
                    # most of why the replay was slower than eager.  Prepare now, as _apply_pending does after every update: the
                    # captured step reads the prepared buffers by address.
                    # Twice, with a warm-up pass in between: the input-gradient convolutions' Winograd / limb images are derived from
                    # the PREPARED transposed weights, so their recipes can only be registered by a pass that already found those.
                    for rnd in range(2):
                        if self.prefold is not None:
                            self.prefold.run()
                        if self.prep is not None:
                            if self.prefold is not None and self.prefold.valid[0]:
                                self.prep.run()
                            else:
                                self.prep.invalidate()
                        if rnd == 0:
                            self._graph_body(st)
                s.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=s):
                    out = self._graph_body(st)
            finally:
                Fn.CAPTURE_ORIGIN = None
            cur.wait_stream(s)
            if len(self._step_graphs) >= 8:
                self._step_graphs.pop(next(iter(self._step_graphs)))
            ent = self._step_graphs[key] = (graph, st, out)
        graph, st, out = ent
        self._fill_static(st, inputs, gts, counts)
        graph.replay()
        vals = out.clone()                   # (the static output is overwritten by the next replay)
        return {k: vals[i] for i, k in enumerate(st.names)}

    @staticmethod
    def _fill_static(st, inputs: Tensor, gts, counts) -> None:
        import numpy as np
        st.x.copy_(inputs, non_blocking=True)
        off = np.zeros(len(counts) + 1, dtype=np.int32)
        off[1:] = np.cumsum(counts)
        if off[-1] > 0:
            dev = st.gb.device
            st.gb[:off[-1]].copy_(torch.cat([g.bboxes.reshape(-1, 4).float() for g in gts], 0).to(dev, non_blocking=True))
            st.gl[:off[-1]].copy_(torch.cat([g.labels.reshape(-1).long() for g in gts], 0).to(dev, non_blocking=True))
        st.goff.copy_(torch.from_numpy(off), non_blocking=True)

    def _forked(self):
        """raw handles of the streams one step forks to and joins again (trailing weight gradients, the second tower stream)"""
        return (Fn.trail_stream(self.device).cuda_stream, Fn.aux_stream(self.device).cuda_stream)

    def _teacher(self, inputs: Tensor, data_samples, it: int):
        """the frozen teacher's half of step `it` on the current (side) stream: eager launches, or one graph replay"""
        if self.teacher_graphs is None:
            return self.model.teacher_pass(inputs, data_samples)
        out = self.teacher_graphs.run(inputs, slot=it)
        gts, _, metas = unpack_gt_instances(data_samples)         # targets depend on the GT: eager
        out.targets = self.model.bbox_head._targets(out.sizes, gts, metas, self.device)
        return out

    PHASES = None       # tools/phase_times.py: a list to which each step appends (name, event) marks on the step's stream

    def _mark(self, name: str) -> None:
        if ERDTrainer.PHASES is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            ERDTrainer.PHASES.append((name, ev))

    def train_step(self, inputs: Tensor, data_samples, next_batch=None) -> Dict[str, Tensor]:
        """`next_batch` = (inputs, data_samples) of the FOLLOWING step when the loader has it (a prefetching loader does):
        the frozen teacher's half of that step is then queued on the side stream behind this step's losses, so it runs
        next to this step's backward pass instead of next to the following forward pass (the teacher depends on no
        parameter this step updates).  The following call must pass that same `inputs` object to pick the result up."""
        model = self.model
        cur = torch.cuda.current_stream(self.device)
        if self.step_graph:
            self._apply_pending()            # the SGD update of the previous step: one eager launch (its learning rate is a host value)
            log_vars = self._train_step_graph(inputs, data_samples)
            if log_vars is not None:
                self._pending = True
                self._pending_lr = self.last_lr = self.lr_at(self.iter, self.epoch_factor)
                self.iter += 1
                return log_vars
        if self.overlap_teacher:
            # The frozen teacher (forward + ERS + NMS) runs on the side stream, concurrently with (a) the tail of the
            # previous step's gradient all-reduce + the deferred SGD launch and (b) the student's forward on the
            # main stream: two independent kernel streams fill each other's partially filled dispatch rounds.
            ahead = self._teacher_ahead
            self._teacher_ahead = None
            if ahead is not None and ahead[0] is inputs:
                teacher_out = ahead[1]                   # queued on the side stream during the previous step's backward
            else:
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side), torch.no_grad():
                    teacher_out = self._teacher(inputs, data_samples, self.iter)
            self._mark("step")
            self._apply_pending()
            self.flat.zero_grad()
            K.zero_arena_begin(self.device, forked_streams=self._forked())
            self._mark("update_done")
            if teacher_out.trunk is not None:      # shared frozen trunk: the student starts at its first trainable stage
                cur.wait_event(teacher_out.trunk_event)
                teacher_out.trunk[0].record_stream(cur)
            with K.distillation_forward(K.WINO_FROZEN_TRUNK):
                s_cls, s_bbox, sizes = model._forward_cat(inputs, trunk=teacher_out.trunk)
            self._mark("student_forward_done")
            cur.wait_stream(self.side)
            for t in teacher_out.tensors():
                t.record_stream(cur)
            self._mark("teacher_joined")
            losses = model.bbox_head.loss_cat(teacher_out.t_cls, teacher_out.t_bbox, s_cls, s_bbox, sizes, data_samples,
                                              teacher_out.ers, teacher_out.keep, model.ori_num_classes,
                                              model.dist_loss_weight, targets=teacher_out.targets)
        else:
            self._apply_pending()
            self.flat.zero_grad()
            K.zero_arena_begin(self.device)
            losses = model(inputs, data_samples, mode="loss")
        total, log_vars = parse_losses(losses)
        self._mark("losses_done")
        if next_batch is not None and self.overlap_teacher:
            # the teacher of the following step: behind everything the side stream holds for this one, next to this backward.
            # `cur` holds whatever PRODUCED next_batch (Runner.train prepares batch t+1 on the current stream before it calls
            # train_step(t): preprocess / resize kernels, H2D copies) plus this step's forward and losses, but not yet its
            # backward -- the side stream must be ordered behind that producer (without this join the teacher could read
            # half-written pixels as soon as the host runs ahead), and still runs next to backward(t).
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side), torch.no_grad():
                self._teacher_ahead = (next_batch[0], self._teacher(next_batch[0], next_batch[1], self.iter + 1))
        self._pending_lr = self.last_lr = self.lr_at(self.iter, self.epoch_factor)      # (per-bucket updates start inside backward)
        if self.sync is not None:
            self.sync.arm()
        total.backward()
        self._mark("backward_chain_done")
        Fn.trail_join(self.device)        # the trailing weight gradients of the backbone (functional._Trail)
        self._mark("trail_joined")
        K.zero_arena_end()
        self._pending = True
        if self.bucket_update:
            self.sync.issue_rest()        # the tail bucket: queued now, joined by the next step (or flush())
        self.iter += 1
        if not self.overlap_teacher:
            self._apply_pending()
        return log_vars
