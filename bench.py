#!/usr/bin/env python3
"""bench.py -- images/sec of the GFL-R50 40+40 ERD incremental training step at 1333x800 on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, see `self_launch`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (the driver's form: one rank per GPU, RANK / WORLD_SIZE from the env)

Other workloads of BASELINE.json (parity-tested configurations; their lines go to profiles/):
    --compute bf16            configs[2], per-GPU leg (bs 4 of the bs-32 job)
    --arch r101_70_10         configs[3], per-GPU leg (bs 2 of the bs-16 job)
    --mixed-res               configs[4]: five landscape shapes 704..800 x 1088..1333, hipGraph-captured teacher

One step = teacher fwd + ERS + NMS + student fwd + the five loss groups + student backward + gradient mean over
ranks (RCCL) + SGD update, on one synthetic batch already resident in HBM (BASELINE.json configs[1]: bs=4 per GPU,
fp32, procedural weights).  Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured with HIP events
on the launch stream over the timed region) and, at N=1, `cpu_baseline` (the oracle restatement timed on the host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL across processes)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

H, W = 800, 1333
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = f32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 (MI355X_MICROARCH.md)
# what the bf16 pipe SUSTAINS on random operands (tools/mfma_peak.hip, profiles/r04_mfma_peak_random.txt: 1 865-1 873 TF at 1.84 GHz;
# constant operands hold 2.39 GHz and 2 472 TF -- power management, not issue rate).  Reported BESIDE the nominal peak, never as `peak`.
BF16_MFMA_SUSTAINED_TFLOPS = 1867.0
FP32_MFMA_SUSTAINED_TFLOPS = 154.2  # (random operands: the fp32 MFMA keeps its clock)
HBM_PEAK_GBS = 8000.0
# algorithmic work per image per step at 800x1344: teacher fwd + student fwd + student bwd
STEP_GFLOP_PER_IMAGE = {"r50_40_40": 1663.6,       # 431.8 + 436.0 + 795.8
                        "r101_70_10": 2299.9}      # 590.9 + 595.1 + 1113.9
# the student's copy of the frozen stem + layer1 that the shared trunk does not execute (2.53 + 14.31 GMAC, both depths)
TRUNK_GFLOP_PER_IMAGE = 33.68
ARCH = {"r50_40_40": dict(student="gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py", teacher="gfl_r50_fpn_1x_coco_first_40_cats.py",
                          c_old=40, c_all=80, batch=4, label="gfl_r50_fpn first_40_incre_last_40 ERD"),
        "r101_70_10": dict(student="gfl_r101_fpn_1x_coco_first_70_incre_last_10_cats.py", teacher="gfl_r101_fpn_1x_coco_first_70_cats.py",
                           c_old=70, c_all=80, batch=2, label="gfl_r101_fpn first_70_incre_last_10 ERD")}
# mixed-resolution batches (BASELINE configs[4]; SURVEY 8(d): the landscape bucket of AspectRatioBatchSampler,
# datasets/samplers/batch_sampler.py:36-49): (h, w) of the batch before padding to /32
MIXED_SHAPES = [(800, 1333), (800, 1216), (800, 1088), (768, 1333), (704, 1333)]


def synthetic_gpu_batch(bs: int, seed: int, device, cfg=None, num_new: int = 40, H: int = H, W: int = W):
    """post-preprocess batch of the reference's demo_mm_inputs shape (mmdet/testing/_utils.py:89-202):
    uint8 pixels -> BGR->RGB -> (x-mean)/std -> zero pad to /32, 1..9 random boxes, labels in [0, C_new)."""
    from erd_amd import DetDataSample, InstanceData
    from erd_amd.synthetic import demo_batch
    if cfg is None:
        from erd_amd import Config
        cfg = Config.fromfile(os.path.join(ROOT, "configs", "gfl_increment", "gfl_r50_fpn_1x_coco_first_40_cats.py"))
    dp = cfg.model.data_preprocessor
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    mean = torch.tensor(list(dp.mean), device=device).view(3, 1, 1)
    std = torch.tensor(list(dp.std), device=device).view(3, 1, 1)
    x = torch.zeros((bs, 3, Hp, Wp), device=device)
    samples = []
    imgs, boxes, labels = demo_batch(bs, H, W, num_new, seed)
    for i in range(bs):
        x[i, :, :H, :W] = (imgs[i].to(device)[[2, 1, 0]].float() - mean) / std
        ds = DetDataSample(metainfo=dict(img_shape=(H, W), pad_shape=(Hp, Wp), batch_input_shape=(Hp, Wp)))
        ds.gt_instances = InstanceData(bboxes=boxes[i].to(device), labels=labels[i].to(device))
        samples.append(ds)
    return x, samples


def build_model(device, rank: int, arch: str = "r50_40_40"):
    """through the reference's own boundary: config files + MODELS.build + teacher checkpoint on disk
    (gfl_increment_erd.py:95-122).  Weights are procedural (erd_amd/synthetic.py; no network for checkpoints)."""
    from erd_amd import Config, MODELS
    from erd_amd.synthetic import procedural_state_dict, state_shapes
    cdir = os.path.join(ROOT, "configs", "gfl_increment")
    cfg = Config.fromfile(os.path.join(cdir, ARCH[arch]["student"]))
    tcfg_file = os.path.join(cdir, ARCH[arch]["teacher"])
    tsd = procedural_state_dict(state_shapes(MODELS.build(Config.fromfile(tcfg_file).model)), seed=0)
    ckpt = os.path.join(tempfile.gettempdir(), f"erd_teacher_{arch}_rank{rank}_{os.getpid()}.pth")
    torch.save(dict(state_dict=tsd), ckpt)
    cfg.model.ori_setting.ori_checkpoint_file = ckpt
    cfg.model.ori_setting.ori_config_file = tcfg_file
    torch.manual_seed(1234)                      # the student's fresh new-class rows: same on every rank
    model = MODELS.build(cfg.model)
    os.remove(ckpt)
    return model.to(device).train(), cfg


def _host_cpu():
    """(model name, physical cores, logical cores) of the host this runs on"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    try:      # a container may expose fewer cores than the machine has
        logical = min(logical, len(os.sched_getaffinity(0)))
        physical = min(physical, logical)
    except AttributeError:
        pass
    return model, physical, logical


def cpu_baseline():
    """the oracle restatement (kind 'port') on the host, batch 2, synthetic demo_mm_inputs-shaped
    images at 1333x800, for (i) BASELINE.json configs[0] -- plain GFL first-40 forward + loss -- and (ii) the ERD step
    (teacher fwd + ERS + NMS + student fwd + losses + backward), the workload of the GPU number next to it.
    Section 4 prescribes all physical cores, median of 3 after 1 warm-up: that is `all_cores`.  On a 128-core host that
    oversubscribes oneDNN (round 2 measured HALF of what 8 container cores gave), so the ERD step is also timed once at
    16 / 32 / 64 threads; `value` / `cores` report the 32-THREAD point -- the count the parity suite validates the oracle at
    (tests/conftest.py) -- and `fastest_of_sweep` / `thread_sweep` carry the rest.
    A bounded sample: 4 + 3 ERD steps and 4 + 1 GFL passes."""
    from oracle import erd_oracle as O
    model, physical, logical = _host_cpu()
    tsd = O.procedural_state_dict(40, seed=0)
    ssd = O.student_state_from_teacher(tsd, 80, seed=1)
    nimg = 2
    imgs, boxes, labels = O.synthetic_batch(nimg, H, W, 40, seed=0)
    x, metas = O.preprocess(imgs)

    def gfl_first40():
        with torch.no_grad():
            cls, bbox = O.gfl_forward(tsd, x)
            return O.parse_losses(O.gfl_head_loss(cls, bbox, boxes, labels, metas, 40))

    def erd_step():
        sd = {k: (v.clone().requires_grad_(True) if O.trainable(k) and v.dtype == torch.float32 else v)
              for k, v in ssd.items()}
        O.parse_losses(O.erd_step_loss(tsd, sd, x, boxes, labels, metas, 40, 80)).backward()

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2], ts

    torch.set_num_threads(physical)
    t_gfl, all_gfl = timed(gfl_first40, 3, 1)
    t_erd, all_erd = timed(erd_step, 3, 1)
    sweep = {physical: t_erd}
    for th in (64, 32, 16):
        if th < physical:
            torch.set_num_threads(th)
            sweep[th] = timed(erd_step, 1, 0)[0]       # (primitives are cached by the all-cores leg: no second warm-up)
    # `value` is quoted at the thread count the parity suite VALIDATES the oracle at (tests/conftest.py: 32; fewer only on a smaller host):
    # at 16 threads torch-CPU's fp32 backward of a stride-2 bottleneck differs from its own 32- / 128-thread result by 4e-3 on the EPYC
    # 9575F of the GPU boxes (tools/dbg/cpu_threads_block_vs_hip.py), so the fastest point of the sweep is reported as a sub-field only.
    checked = min(32, physical)
    if checked not in sweep:
        torch.set_num_threads(checked)
        sweep[checked] = timed(erd_step, 1, 0)[0]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(checked)
    t_gfl_chk = timed(gfl_first40, 1, 0)[0] if checked != physical else t_gfl
    torch.set_num_threads(physical)
    pad = f"{(H + 31) // 32 * 32}x{(W + 31) // 32 * 32}"
    return dict(value=round(nimg / sweep[checked], 4), unit="images/sec", cores=checked, kind="port",
                cpu_model=model, physical_cores=physical, logical_cpus=logical,
                sample=f"batch {nimg} at {pad}: ERD step (teacher fwd+ERS+NMS+student fwd+losses+backward) {sweep[checked]:.2f} s on "
                       f"{checked} threads = the thread count the parity tests check this oracle at (one timed step after the all-cores "
                       f"warm-up; all cores: median of 3 after 1 warm-up); oracle/erd_oracle.py on torch-CPU fp32",
                all_cores={"value": round(nimg / t_erd, 4), "unit": "images/sec", "cores": physical, "seconds": round(t_erd, 3),
                           "form": "BASELINE.md section 4: all physical cores, median of 3 after 1 warm-up"},
                thread_sweep={str(th): {"seconds": round(t, 3), "images_per_sec": round(nimg / t, 4)} for th, t in sorted(sweep.items())},
                fastest_of_sweep={"cores": best, "value": round(nimg / sweep[best], 4),
                                  "note": "a TIME only: below 32 threads the oracle's gradients are not the ones the suite validated"},
                configs0_gfl_first40_fwd_loss={"value": round(nimg / t_gfl_chk, 4), "unit": "images/sec", "cores": checked,
                                               "all_cores": {"value": round(nimg / t_gfl, 4), "seconds": round(t_gfl, 3)}},
                seconds_all={"erd_step": [round(t, 3) for t in all_erd], "gfl_first40": [round(t, 3) for t in all_gfl]})


def _pmc_file(suffix: str, compute: str):
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith(suffix) and (("bf16" in f) == (compute == "bf16"))) \
        if os.path.isdir(pdir) else []
    return os.path.join(pdir, files[-1]) if files else None


def pmc_provenance(compute: str = "f32") -> dict:
    """The committed PMC summaries are STATIC (counters cannot be read from inside this process): each records the sha256 of the
    kernel sources of the library it was taken on (tools/pmc_traffic.py / pmc_mfma.py `_meta.csrc_sha256` = erd_csrc_sha()); next to
    it the live library's -- `pmc_stale` says the counters describe other code than the one benched."""
    from erd_amd import _lib
    live = _lib.load().erd_csrc_sha().decode()
    shas = {}
    for suffix in ("pmc_traffic.json", "pmc_mfma_busy.json"):
        f = _pmc_file(suffix, compute)
        if f is not None:
            shas["profiles/" + os.path.basename(f)] = json.load(open(f)).get("_meta", {}).get("csrc_sha256")
    return {"pmc_source_sha": shas, "library_csrc_sha": live, "pmc_stale": (not shas) or any(v != live for v in shas.values()),
            "compiler": _hipcc_version()}


def _hipcc_version() -> str:
    """the compiler the image would build the library with (the source sha covers sources and flags -- csrc/Makefile -- not the compiler)"""
    import subprocess
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True, timeout=30).stdout
        lines = [l.strip() for l in out.splitlines() if l.strip()]
        return "; ".join(l for l in lines if l.startswith(("HIP version", "AMD clang version")))[:200] or (lines[0][:200] if lines else "unknown")
    except Exception:
        return "unknown"


def pmc_traffic_per_launch(symbol_prefix: str, compute: str = "f32"):
    """HBM-side bytes per launch of the kernels whose symbol starts with `symbol_prefix`, from the committed PMC
    summary (two separate rocprofv3 --pmc passes, tools/pmc_traffic.py; FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process -> None when absent."""
    f = _pmc_file("pmc_traffic.json", compute)
    if f is None:
        return None, None
    d = json.load(open(f))
    n = sum(v["launches"] for k, v in d.items() if k.startswith(symbol_prefix) and v.get("traffic_MB"))
    if not n:
        return None, None
    mb = sum(v["launches"] * v["traffic_MB"] for k, v in d.items() if k.startswith(symbol_prefix) and v.get("traffic_MB"))
    return int(mb / n * 1e6), "profiles/" + os.path.basename(f)


def pmc_mfma_busy(symbol_prefix: str, compute: str = "f32"):
    """matrix-pipe busy fraction of the kernels whose symbol starts with `symbol_prefix` from the committed PMC summary
    (one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass, tools/pmc_mfma.py); None when absent"""
    f = _pmc_file("pmc_mfma_busy.json", compute)
    if f is None:
        return None, None
    d = json.load(open(f))
    rows = [(v["launches"], v["mfma_busy_fraction"]) for k, v in d.items() if k.startswith(symbol_prefix)]
    n = sum(r[0] for r in rows)
    return (round(sum(a * b for a, b in rows) / n, 4), "profiles/" + os.path.basename(f)) if n else (None, None)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, as the reference's launcher
    does (tools/dist_train.sh:11-19 -> torch.distributed.launch; configs/_base_/default_runtime.py:14 backend 'nccl').
    The parent never touches the GPU (no HIP call, no torch.cuda.is_available()) and never exec()s: it starts
    `python -m torch.distributed.run ... bench.py <same flags>` as a CHILD process, relays rank 0's JSON line and exits
    with the children's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_args, skip = [], False
    for a in argv:                     # drop `--launcher X` / `--launcher=X` by position (never by value); the children get `--launcher none`
        if skip:
            skip = False
        elif a == "--launcher":
            skip = True
        elif not a.startswith("--launcher="):
            child_args.append(a)
    child_args += ["--launcher", "none"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ERD_BENCH_CHILD="1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:          # rank 0's JSON line (and anything else the ranks print) goes through unchanged
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# timing classes of erd_amd.kernels -> the kernel SYMBOL they run on (rocprof's unit); a roofline is a kernel's
SYMBOLS = {"wino_conv_kernel": ("conv_wino_fwd", "conv_wino_dgrad"),
           "conv_igemm_kernel": ("conv_igemm_fwd", "conv_igemm_dgrad"),
           "conv_thin_x3_kernel": ("conv_thin_fwd", "conv_thin_dgrad"),
           "conv_wgrad_row3_kernel": ("conv_wgrad_row3",),
           "conv_wgrad_kernel": ("conv_wgrad",)}
WINO_EXECUTED = 16.0 / 36.0       # F(2x2,3x3) runs 16 of the 36 multiplications a direct 3x3 convolution counts per 2x2 outputs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: BASELINE's -- r50_40_40: 4, r101_70_10: 2)")
    ap.add_argument("--arch", choices=sorted(ARCH), default="r50_40_40",
                    help="r50_40_40: BASELINE configs[1] (the headline).  r101_70_10: configs[3] (deeper backbone, 70 old + 10 new classes)")
    ap.add_argument("--mixed-res", action="store_true",
                    help="BASELINE configs[4]: every batch draws its (h, w) from the five landscape shapes of SURVEY 8(d); the frozen "
                         "teacher is replayed from one hipGraph per padded shape (implies --teacher-graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-strict-fp32", action="store_true",
                    help="skip the short `--compute f32` leg whose rate the f32x3 line carries as `strict_fp32`")
    ap.add_argument("--teacher-graph", action="store_true",
                    help="replay the frozen teacher's pass from a hipGraph (BASELINE configs[4]); same arithmetic")
    ap.add_argument("--compute", choices=["f32x3", "f32", "bf16"], default="f32x3",
                    help="f32x3 (default; BASELINE configs[1], the headline): fp32 maps, fp32 accumulation, fp32 results; the direct "
                         "implicit-GEMM launches form each fp32 product on the bf16 matrix cores from exact three-limb splits of both "
                         "multiplicands (round-to-nearest limbs: what is dropped is zero-mean and below 2^-23 of the product; gfx950's fp32 MFMA runs at 1/16 of the bf16 rate), so do "
                         "the weight-gradient launches and the transform-domain products of the Winograd launches.  f32: every launch on the fp32 matrix cores "
                         "(the A/B sibling).  bf16: BASELINE configs[2] -- the 1x1 / 3x3 convolutions on the bf16 matrix cores with fp32 "
                         "accumulation, feature maps and their gradients STORED as bf16; head outputs, statistics, losses, parameters "
                         "and parameter gradients stay fp32")
    ap.add_argument("--step-graph", action="store_true",
                    help="replay the whole step (everything between two SGD updates) from one hipGraph; same arithmetic")
    ap.add_argument("--no-teacher-ahead", action="store_true",
                    help="do not hand the trainer the following batch (teacher of step t+1 then runs next to the forward of step t+1)")
    ap.add_argument("--no-streamk", action="store_true",
                    help="A/B aid: tile-parallel implicit-GEMM launches instead of the stream-K split (every world size uses "
                         "stream-K by default, so the N = 1 point of a scaling curve is the sibling of the N > 1 points)")
    ap.add_argument("--serial", action="store_true",
                    help="no stream concurrency in the timed region either (the rocprofv3 companion run)")
    ap.add_argument("--occupy-cus", type=int, default=0,
                    help="A/B aid, never the headline: a spin kernel (tools/occupy_cus.hip, compiled into /tmp) holds this many CUs on a side "
                         "stream for the whole timed region -- a single-GPU stand-in for RCCL channels resident beside the backward pass "
                         "(no N > 1 hardware reaches the builder: tools/cu_theft_step.sh, profiles/r05_cu_theft_step.txt)")
    ap.add_argument("--launcher", choices=["auto", "spawn", "none"], default="auto",
                    help="auto: --gpus N > 1 without a torchrun environment starts its own N ranks (torch.distributed.run as a child "
                         "process).  spawn: do that at any N (the world-1 test of the path).  none: never")
    args = ap.parse_args()

    # ---- BEFORE anything touches the GPU: are we the launcher?  (the driver's torchrun form sets WORLD_SIZE; a plain
    # `python bench.py --gpus 8` does not)
    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.launcher != "none" and not in_torchrun and os.environ.get("ERD_BENCH_CHILD") != "1" and \
            (args.launcher == "spawn" or args.gpus > 1):
        raise SystemExit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run, or let bench.py start "
                         f"its own ranks (--launcher auto, the default)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from erd_amd.dist_utils import backend_name, device_index
    dev_index = device_index(local_rank)       # (ERD_DIST_BACKEND=gloo: ranks may share a GPU -- the world-2 correctness vehicle)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or in_torchrun or os.environ.get("ERD_FORCE_DIST") == "1":      # (world 1 under torchrun: the RCCL path on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend_name(), rank=rank, world_size=world)   # 'nccl' == RCCL on ROCm

    from erd_amd import functional as Fn
    from erd_amd import kernels as K
    from erd_amd.engine import ERDTrainer
    K.set_compute(args.compute)
    if args.no_streamk:
        K.STREAMK = False
    arch = ARCH[args.arch]
    if args.batch is None:
        args.batch = arch["batch"]
    if args.mixed_res:
        args.teacher_graph = True
    model, cfg = build_model(device, rank, args.arch)
    opt = cfg.optim_wrapper.optimizer
    trainer = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                         base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=args.batch,
                         auto_scale_lr=cfg.auto_scale_lr.enable, teacher_graph=args.teacher_graph,
                         step_graph=args.step_graph)
    num_new = arch["c_all"] - arch["c_old"]
    shapes = MIXED_SHAPES if args.mixed_res else [(H, W)]
    # two batches per shape; a mixed-resolution run walks the shapes round-robin (every shape equally often)
    batches = [synthetic_gpu_batch(args.batch, seed=rank * 1000 + i, device=device, cfg=cfg, num_new=num_new,
                                   H=shapes[i % len(shapes)][0], W=shapes[i % len(shapes)][1]) for i in range(2 * len(shapes))]
    pad32 = lambda v: (v + 31) // 32 * 32
    area = lambda j: pad32(shapes[j % len(shapes)][0]) * pad32(shapes[j % len(shapes)][1]) / float(pad32(H) * pad32(W))

    graph_mode = trainer.step_graph
    trail_mode = Fn.WGRAD_TRAIL

    def set_serial(flag: bool):
        """serial = one HIP stream, kernels back to back: per-launch durations are then well defined"""
        trainer.flush()
        trainer.overlap_teacher = (not flag) and trainer.is_erd
        trainer.step_graph = graph_mode and not flag          # (per-launch events cannot be taken inside a replayed graph)
        Fn.TOWERS_ON_TWO_STREAMS = not flag
        Fn.WGRAD_TRAIL = trail_mode and not flag              # (trailing weight gradients overlap the input-gradient chain)
    if args.serial:
        set_serial(True)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    log = None
    # the loader hands the trainer the FOLLOWING batch as well (Runner.train does the same with a one-batch look-ahead): the
    # frozen teacher's half of step t+1 is queued next to the backward pass of step t.  Every timed step still runs one
    # teacher pass, one student forward / backward and one update (+0.65 % fp32, +6 % bf16; --no-teacher-ahead: off)
    ahead = not args.no_teacher_ahead and not args.serial
    seq = lambda j: batches[j % len(batches)]                    # one batch sequence across warm-up and timed steps, so that
    nb = lambda j: seq(j + 1) if ahead else None                 # the batch announced by the last warm-up step IS the first timed one
    warm = max(args.warmup, len(batches)) if args.mixed_res else args.warmup     # (every shape's graph is captured before the clock starts)
    # data parallel: how many CUs do the whole-chip grids leave to RCCL's resident kernels?  Probed with the process group live, all ranks
    # adopt one value (ERDTrainer.tune_cu_reserve); N = 1 is untouched (no probe, reserve 0).  ERD_CU_RESERVE=<n> pins it.
    cu_reserve = {"cu_reserve": 0, "probed": False}
    if dist.is_initialized() and dist.get_world_size() > 1 and not args.mixed_res:
        if os.environ.get("ERD_CU_RESERVE"):
            K.set_cu_reserve(int(os.environ["ERD_CU_RESERVE"]))
            trainer.cu_reserve = int(os.environ["ERD_CU_RESERVE"])
            cu_reserve = {"cu_reserve": trainer.cu_reserve, "probed": False, "note": "pinned by ERD_CU_RESERVE"}
        else:
            cu_reserve = trainer.tune_cu_reserve(batches)
    for j in range(warm):
        log = trainer.train_step(*seq(j), next_batch=nb(j))
    trainer.flush()
    barrier()
    occupier = None
    if args.occupy_cus > 0:
        import ctypes
        import subprocess
        import tempfile
        so = os.path.join(tempfile.mkdtemp(prefix="erd_occupy_"), "occupy_cus.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "occupy_cus.hip"),
                        "-o", so], check=True)
        occ = ctypes.CDLL(so)
        occ.occupy_cus.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
        occupier = (torch.cuda.Stream(device=device), torch.zeros(1, dtype=torch.int64, device=device))
        torch.cuda.synchronize()
        # two workgroups of 1024 threads + 64 KB of LDS fill a CU; long enough for the timed region at half the usual rate
        occ.occupy_cus(2 * args.occupy_cus, int(args.steps * 0.09 * 2.4e9), occupier[1].data_ptr(), occupier[0].cuda_stream)

        def barrier():      # the ranks still rendezvous; the device wait is the trainer's streams joined into the current one
            if dist.is_initialized():      # (a device-wide wait would sit out the spin kernel)
                dist.barrier()
            trainer.join_streams()
            torch.cuda.current_stream(device).synchronize()
    t0 = time.perf_counter()
    for j in range(warm, warm + args.steps):
        log = trainer.train_step(*seq(j), next_batch=nb(j))
    trainer.flush()                   # the deferred SGD of the last step belongs to the timed region
    barrier()
    dt = time.perf_counter() - t0
    if occupier is not None:
        still_held = not occupier[0].query()            # must be True: the spin kernel outlived the timed region
        torch.cuda.synchronize()
        if not still_held:
            raise SystemExit("bench.py --occupy-cus: the spin kernel ended inside the timed region -- the measurement is void "
                             "(raise the spin length in bench.py for this --steps)")
    rel_area = sum(area(j) for j in range(warm, warm + args.steps)) / args.steps     # mean padded area of the timed steps / 800x1344
    # roofline leg: the same steps again with HIP events around every GEMM-shaped launch, streams serialized
    # (overlapping kernels have no well-defined individual duration).  Not part of `value`.
    ktime, rsteps = None, 0
    if not args.no_kernel_timing:
        set_serial(True)
        trainer.train_step(*batches[0])
        trainer.flush()
        torch.cuda.synchronize()
        rsteps = min(args.steps, 4)
        K.timing_begin()
        for i in range(rsteps):
            trainer.train_step(*batches[i % (2 if not args.mixed_res else len(batches))])
        trainer.flush()
        ktime = K.timing_end()
        set_serial(args.serial)
    # ---- the strict-fp32 sibling of the headline: the same step with EVERY GEMM-shaped launch on the fp32 matrix cores
    # (`--compute f32`: IEEE fp32 products, the reference's arithmetic, resnet.py:268-300 -> ATen conv), a short leg on a second
    # trainer built under that mode; reported next to `value`, never as `value`
    strict = None
    if args.compute == "f32x3" and not args.no_strict_fp32 and not args.serial and not args.mixed_res:
        K.set_compute("f32")
        try:
            model2, _ = build_model(device, rank, args.arch)
            tr2 = ERDTrainer(model2, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                             base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=args.batch,
                             auto_scale_lr=cfg.auto_scale_lr.enable, teacher_graph=args.teacher_graph, step_graph=args.step_graph)
            s_warm, s_steps = 2, 5
            for j in range(s_warm):
                tr2.train_step(*seq(j), next_batch=nb(j))
            tr2.flush()
            barrier()
            t1 = time.perf_counter()
            for j in range(s_warm, s_warm + s_steps):
                slog = tr2.train_step(*seq(j), next_batch=nb(j))
            tr2.flush()
            barrier()
            dt2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            if dist.is_initialized():
                dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
            strict = {"value": round(args.batch * world * s_steps / float(dt2.item()), 3), "unit": "images/sec",
                      "ms_per_step": round(1e3 * float(dt2.item()) / s_steps, 3), "steps": s_steps, "warmup": s_warm,
                      "compute_mode": "f32", "loss": round(float(slog["loss"].detach()), 6),
                      "arithmetic": "fp32 throughout, every GEMM-shaped launch on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)"}
            del tr2, model2
        finally:
            K.set_compute(args.compute)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    devices = [f"rank {rank}: cuda:{dev_index} {torch.cuda.get_device_name(dev_index)}"]
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    dt = float(tmax.item())
    loss = float(log["loss"].detach()) if log is not None else float("nan")

    if rank == 0:
        images = args.batch * world * args.steps
        prec = "bf16" if args.compute == "bf16" else "f32"
        which = {("r50_40_40", "f32", False): "BASELINE configs[1]", ("r50_40_40", "bf16", False): "BASELINE configs[2], per-GPU leg",
                 ("r101_70_10", "f32", False): "BASELINE configs[3], per-GPU leg, fp32", ("r101_70_10", "bf16", False): "BASELINE configs[3], per-GPU leg, bf16",
                 ("r50_40_40", "f32", True): "BASELINE configs[4], per-GPU leg, fp32", ("r50_40_40", "bf16", True): "BASELINE configs[4], per-GPU leg, bf16"}.get(
                     (args.arch, prec, args.mixed_res), "a combination BASELINE.json does not name")
        arithmetic = {"f32x3": "fp32 maps / accumulation / results; the products of the direct implicit-GEMM and weight-gradient launches are formed "
                               "on the bf16 matrix cores from exact three-limb splits of both fp32 multiplicands (6 of 9 limb products, round-to-nearest limbs: dropped "
                               "part zero-mean and < 2^-23 of a product); so are the transform-domain products of the Winograd F(2x2,3x3) launches" +
                               ("" if K.wino_x3() else " -- EXCEPT here: ERD_WINO_X3=0 keeps the Winograd launches on the fp32 matrix cores"),
                      "f32": "fp32 throughout, every GEMM-shaped launch on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
                      "bf16": "bf16 matrix cores, bf16-stored maps, fp32 accumulate / statistics / losses"}[args.compute]
        res = "mixed resolution " + "/".join(f"{h}x{w}" for h, w in MIXED_SHAPES) + " (round-robin, each padded to /32)" if args.mixed_res \
            else "1333x800 padded to 800x1344"
        out = {
            "metric": f"images/sec GFL-{'R50 40+40' if args.arch == 'r50_40_40' else 'R101 70+10'} incre step @1333x800",
            "value": round(images / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": warm, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": prec, "arithmetic": arithmetic, "compute_mode": args.compute, "data": "synthetic",
            "config": {"workload": f"{arch['label']} ({which}), {res}, " +
                                   ("fp32" if prec == "f32" else ("bf16 matrix cores, bf16-stored maps, fp32 accumulate / statistics / losses"
                                                                 if K.BF16_STORAGE else "bf16 multiplicands / fp32 accumulate+storage")) +
                                   ", procedural weights", "batch_per_gpu": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "loss": round(loss, 6), "streams": "serial" if args.serial else ("teacher(t+1)||backward(t), cls||reg towers, trailing weight gradients" if ahead
                                                             else "teacher||student, cls||reg towers, trailing weight gradients"),
            "teacher": "hipGraph replay (one graph per padded shape and buffer parity)" if args.teacher_graph else "eager launches",
            "step_graph": bool(trainer.step_graph),
        }
        if occupier is not None:
            out["occupied_cus"] = {"cus": args.occupy_cus, "held_for_the_whole_timed_region": bool(still_held),
                                   "note": "A/B aid (tools/occupy_cus.hip): NOT the headline configuration"}
        out["collectives"] = {"backend": ({"nccl": "nccl (RCCL)", "gloo": "gloo (host-staged: correctness vehicle, not a performance path)"}[dist.get_backend()]
                                          if dist.is_initialized() else None),
                              "world_size": dist.get_world_size() if dist.is_initialized() else 1, "devices": devices,
                              "cu_reserve": cu_reserve,
                              "launched_by": "bench.py self_launch -> torch.distributed.run" if os.environ.get("ERD_BENCH_CHILD") == "1"
                              else ("torch.distributed.run" if in_torchrun else "single process")}
        out["kernel_config"] = "stream-K implicit GEMM" if (K.STREAMK and (K.STREAMK_MULTIRANK or not K._multi_rank())) \
            else "tile-parallel implicit GEMM"
        shared = bool(getattr(model, "shares_trunk", lambda: False)())
        # student and teacher hold the same frozen stem + layer1: computed once per step and fed to both.  The skipped
        # launches are the student's copy (2.53 + 14.31 GMAC per image, BASELINE.md section 3 / SURVEY Appendix A);
        # `roofline.x_fp32_mfma_ceiling` keeps counting the ALGORITHMIC work of the reference's step (both copies)
        out["shared_frozen_trunk"] = {"enabled": shared, "skipped_gflop_per_image": TRUNK_GFLOP_PER_IMAGE if shared else 0.0}
        if ktime:
            peak_tf = BF16_MFMA_PEAK_TFLOPS if args.compute == "bf16" else FP32_MFMA_PEAK_TFLOPS   # every GEMM class follows --compute
            # ---- the dominant KERNEL SYMBOL (rocprof's unit: the Winograd kernel's forward and input-gradient launches are
            # one symbol) and its roofline on the flops it EXECUTES
            wino_on_bf16 = K.wino_x3()       # "f32x3": the Winograd launches run as wino_x3_kernel (six bf16 limb products per transform-domain product)
            symbols = dict(SYMBOLS)
            if args.compute == "bf16":      # the bf16 mode's thin 1x1 launches run on their own kernel (conv_thin.hip, round 6)
                symbols["conv_thin_bf16_kernel"] = symbols.pop("conv_thin_x3_kernel")
            if wino_on_bf16:     # two symbols: items of 128 output channels (wino_x3p_kernel; the launch picks, kernels.wino_conv3x3 names it) and of 64
                symbols["wino_x3_kernel"] = symbols.pop("wino_conv_kernel")
                symbols["wino_x3p_kernel"] = ("conv_wino_fwd_p", "conv_wino_dgrad_p")
            wino_syms = ("wino_x3_kernel", "wino_x3p_kernel") if wino_on_bf16 else ("wino_conv_kernel",)
            groups = {}
            for sym, classes in symbols.items():
                rs = [ktime[c] for c in classes if c in ktime]
                if rs:
                    groups[sym] = dict(ms=sum(r["ms"] for r in rs), flop=sum(r["flop"] for r in rs), launches=sum(r["launches"] for r in rs),
                                       min_bytes=sum(r["min_bytes"] for r in rs))
            sym, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
            # executed flops per algorithmic flop, and the matrix pipe they run on: Winograd 16/36 on the fp32 pipe; the three-limb
            # form of the direct launches 6 bf16 MFMA flops per fp32 flop on the bf16 pipe
            x3 = args.compute == "f32x3"
            X3_SYMS = ("conv_igemm_kernel", "conv_thin_x3_kernel", "conv_wgrad_row3_kernel", "conv_wgrad_kernel")      # launch classes that run in the three-limb form
            execf = WINO_EXECUTED * (6.0 if wino_on_bf16 else 1.0) if sym in wino_syms else (6.0 if (x3 and sym in X3_SYMS) else 1.0)
            if (x3 and sym in X3_SYMS) or (sym in wino_syms and wino_on_bf16):
                peak_tf = BF16_MFMA_PEAK_TFLOPS
            alg_tf = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
            # PMC counters cannot be read from inside this process: `traffic` / `mfma_busy_pmc` are STATIC values from the
            # committed profile of the same command (fp32 only; the profile names its commit) -- pointers, not measurements
            # (rocprof's symbol of the class: the three-limb weight gradients are instantiations of one kernel template)
            pmc_sym = {"conv_wgrad_row3_kernel": "conv_wgrad_row3_x3_kernel<2, 1", "conv_wgrad_kernel": "conv_wgrad_row3_x3_kernel<2, 2"}.get(sym, sym) \
                if x3 else sym
            traffic, traffic_src = pmc_traffic_per_launch(pmc_sym, args.compute)
            out["roofline"] = {"bound": "mfma", "kernel": sym, "classes": [c for c in symbols[sym] if c in ktime],
                               "achieved": round(alg_tf * execf, 2), "peak": peak_tf, "unit": "TFLOP/s",
                               "frac": round(alg_tf * execf / peak_tf, 4),
                               "basis": "flops the kernel executes on the matrix cores" + (
                                   " (Winograd F(2x2,3x3): 16/36 of the direct-convolution count, x 6 bf16 limb products per product, against the bf16 MFMA peak)"
                                   if (sym in wino_syms and wino_on_bf16) else
                                   " (Winograd F(2x2,3x3): 16/36 of the direct-convolution count)" if execf < 1 else (
                                       " (six bf16 limb products per fp32 product, against the bf16 MFMA peak)" if execf > 1 else " (= the direct-convolution count)")),
                               "algorithmic_tflops": round(alg_tf, 2), "executed_flop_fraction": round(execf, 4),
                               "traffic": traffic, "traffic_unit": "bytes/launch (L2<->fabric, PMC)",
                               "traffic_source": traffic_src, "traffic_static": True,
                               "pass": f"{rsteps} extra steps, streams serialized",
                               "launches_per_step": dom["launches"] // rsteps,
                               "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                               "gflop_per_launch": round(dom["flop"] / dom["launches"] / 1e9, 3),
                               "algorithmic_bytes_per_launch": int(dom["min_bytes"] / dom["launches"])}
            out["roofline"]["mfma_busy_pmc"], out["roofline"]["mfma_busy_source"] = pmc_mfma_busy(pmc_sym, args.compute)
            out["roofline"]["mfma_busy_static"] = True
            out["roofline"].update(pmc_provenance(args.compute))
            # ---- every GEMM-shaped kernel symbol against BOTH of its bounds, on the pipe it runs on: live HIP-event
            # numbers of this run + the committed PMC passes (static, labelled); everything needed to recompute a fraction is in the row
            per_kernel = {}
            for ksym, g in groups.items():
                on_bf16 = args.compute == "bf16" or (x3 and ksym in X3_SYMS) or (ksym in wino_syms and wino_on_bf16)
                k_exec = WINO_EXECUTED * (6.0 if wino_on_bf16 else 1.0) if ksym in wino_syms else (6.0 if (x3 and ksym in X3_SYMS) else 1.0)
                k_peak = BF16_MFMA_PEAK_TFLOPS if on_bf16 else FP32_MFMA_PEAK_TFLOPS
                k_alg = g["flop"] / (g["ms"] * 1e-3) / 1e12
                k_us = 1e3 * g["ms"] / g["launches"]
                k_bytes = g["min_bytes"] / g["launches"]
                k_pmc = ({"conv_wgrad_row3_kernel": "conv_wgrad_row3_x3_kernel<2, 1", "conv_wgrad_kernel": "conv_wgrad_row3_x3_kernel<2, 2"}.get(ksym, ksym)
                         if x3 else ksym)
                k_traffic, k_src = pmc_traffic_per_launch(k_pmc, args.compute)
                k_busy, _ = pmc_mfma_busy(k_pmc, args.compute)
                mfma_frac = k_alg * k_exec / k_peak
                hbm_frac = k_bytes / (k_us * 1e-6) / (HBM_PEAK_GBS * 1e9)
                per_kernel[ksym] = {
                    "pipe": "bf16 MFMA (v_mfma_f32_32x32x16_bf16)" if on_bf16 else "fp32 MFMA",
                    "pipe_peak_tflops": k_peak, "ms_per_step": round(g["ms"] / rsteps, 3), "launches_per_step": g["launches"] // rsteps,
                    "avg_launch_us": round(k_us, 2), "algorithmic_tflops": round(k_alg, 2), "executed_flop_fraction": round(k_exec, 4),
                    "executed_tflops": round(k_alg * k_exec, 2), "mfma_frac": round(mfma_frac, 4),
                    "pipe_sustained_tflops_random_operands": BF16_MFMA_SUSTAINED_TFLOPS if on_bf16 else FP32_MFMA_SUSTAINED_TFLOPS,
                    "mfma_frac_of_sustained": round(k_alg * k_exec / (BF16_MFMA_SUSTAINED_TFLOPS if on_bf16 else FP32_MFMA_SUSTAINED_TFLOPS), 4),
                    "algorithmic_bytes_per_launch": int(k_bytes), "hbm_GBps_algorithmic": round(k_bytes / (k_us * 1e-6) / 1e9, 1),
                    "hbm_frac": round(hbm_frac, 4), "bound": "mfma" if mfma_frac >= hbm_frac else "hbm",
                    "frac_of_binding_roofline": round(max(mfma_frac, hbm_frac), 4),
                    "pmc_traffic_bytes_per_launch": k_traffic, "pmc_traffic_over_algorithmic": round(k_traffic / k_bytes, 2) if k_traffic else None,
                    "pmc_mfma_busy": k_busy, "pmc_static": True, "pmc_source": k_src}
            out["roofline"]["per_kernel"] = per_kernel
            out["roofline"]["sustained_peaks_source"] = "profiles/r04_mfma_peak_random.txt (tools/mfma_peak.hip, random operands)"
            # ---- step level, over the un-instrumented step time of the timed region.  The ROOFLINE FRACTION of the step is
            #  mfma_executed_frac  the share of the step the matrix pipes MUST be busy: sum over the launch classes of the flops they
            #                      execute (Winograd launches at 16/36, three-limb launches 6 x their flops) / the peak of the pipe they
            #                      run on, over the step time.
            # The two other numbers are NOT fractions of a roofline: in the default "f32x3" mode no GEMM-shaped
            # launch runs on the fp32 matrix cores any more, so the step can -- and does -- exceed what they could deliver at peak:
            #  x_fp32_mfma_ceiling           the ALGORITHMIC work of the reference's step (both copies of the frozen
            #                                trunk) per second, as a MULTIPLE of the fp32-MFMA peak (bf16 mode: of the bf16 peak)
            #  x_fp32_mfma_ceiling_executed  the same minus the student's copy of the shared frozen trunk, which this build does not execute
            step_peak = BF16_MFMA_PEAK_TFLOPS if args.compute == "bf16" else FP32_MFMA_PEAK_TFLOPS
            g_img = STEP_GFLOP_PER_IMAGE[args.arch] * rel_area
            step_s = dt / args.steps
            skipped = (TRUNK_GFLOP_PER_IMAGE if shared else 0.0) * rel_area
            per_step = lambda classes: sum(ktime[c]["flop"] for c in classes if c in ktime) / rsteps / 1e9     # GFLOP per step (rank 0's batch)
            wino_alg = per_step(SYMBOLS["wino_conv_kernel"] + ("conv_wino_fwd_p", "conv_wino_dgrad_p"))
            igemm_alg = sum(per_step(SYMBOLS[k_]) for k_ in X3_SYMS) if x3 else 0.0      # (every three-limb class, not only the implicit GEMM)
            exec_gflop_step = args.batch * (g_img - skipped) - wino_alg * (1.0 - WINO_EXECUTED)
            wino_exec = wino_alg * WINO_EXECUTED                                          # Winograd-executed GFLOP per step
            if wino_on_bf16:     # ... which run as six limb products on the bf16 pipe
                pipe_s = ((exec_gflop_step - igemm_alg - wino_exec) / step_peak + (igemm_alg + wino_exec) * 6.0 / BF16_MFMA_PEAK_TFLOPS) * 1e-3
            else:
                pipe_s = ((exec_gflop_step - igemm_alg) / step_peak + igemm_alg * 6.0 / BF16_MFMA_PEAK_TFLOPS) * 1e-3
            out["roofline"]["step_gflop_per_image"] = round(g_img, 1)
            out["roofline"]["step_tflops"] = round(args.batch * g_img / step_s / 1e3, 2)
            out["roofline"]["x_fp32_mfma_ceiling"] = round(out["roofline"]["step_tflops"] / step_peak, 4)
            out["roofline"]["x_fp32_mfma_ceiling_executed"] = round(args.batch * (g_img - skipped) / step_s / 1e3 / step_peak, 4)
            out["roofline"]["x_ceiling_note"] = "multiples of the fp32-MFMA peak, not roofline fractions: the step's roofline fraction is mfma_executed_frac"
            out["roofline"]["mfma_executed_frac"] = round(pipe_s / step_s, 4)
            out["roofline"]["mfma_executed_gflop_per_step"] = round(exec_gflop_step, 1)
            if x3:
                out["roofline"]["three_limb_gflop_per_step"] = round(igemm_alg + (wino_exec if wino_on_bf16 else 0.0), 1)
                out["roofline"]["winograd_on_bf16_pipe"] = bool(wino_on_bf16)
            out["kernels"] = {k: {"ms_per_step": round(r["ms"] / rsteps, 3),
                                  "tflops": round(r["flop"] / (r["ms"] * 1e-3) / 1e12, 2) if r["flop"] else None,
                                  "launches_per_step": r["launches"] // rsteps} for k, r in ktime.items()}
        if strict is not None:
            out["strict_fp32"] = strict
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
