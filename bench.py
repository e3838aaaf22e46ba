#!/usr/bin/env python3
"""bench.py -- images/sec of the GFL-R50 40+40 ERD incremental training step at 1333x800 on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

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
HBM_PEAK_GBS = 8000.0
STEP_GFLOP_PER_IMAGE = 1663.6       # teacher fwd 431.8 + student fwd 436.0 + student bwd 795.8


def synthetic_gpu_batch(bs: int, seed: int, device, cfg=None, num_new: int = 40):
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


def build_model(device, rank: int):
    """through the reference's own boundary: config files + MODELS.build + teacher checkpoint on disk
    (gfl_increment_erd.py:95-122).  Weights are procedural (erd_amd/synthetic.py; no network for checkpoints)."""
    from erd_amd import Config, MODELS
    from erd_amd.synthetic import procedural_state_dict, state_shapes
    cdir = os.path.join(ROOT, "configs", "gfl_increment")
    cfg = Config.fromfile(os.path.join(cdir, "gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py"))
    tcfg_file = os.path.join(cdir, "gfl_r50_fpn_1x_coco_first_40_cats.py")
    tsd = procedural_state_dict(state_shapes(MODELS.build(Config.fromfile(tcfg_file).model)), seed=0)
    ckpt = os.path.join(tempfile.gettempdir(), f"erd_teacher_first40_rank{rank}.pth")
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
    """the oracle restatement (kind 'port') on the host's physical cores, batch 2, synthetic
    demo_mm_inputs-shaped images at 1333x800, median of 3 after 1 warm-up, for (i) BASELINE.json configs[0] -- plain
    GFL first-40 forward + loss -- and (ii) the ERD step (teacher fwd + ERS + NMS + student fwd + losses + backward).
    `value` is (ii), the workload of the GPU number next to it.  A bounded sample: 8 CPU steps, roughly 30-60 s."""
    from oracle import erd_oracle as O
    model, physical, logical = _host_cpu()
    torch.set_num_threads(physical)
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

    def median3(fn):
        fn()                                     # warm-up
        ts = []
        for _ in range(3):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)[1], ts

    t_gfl, all_gfl = median3(gfl_first40)
    t_erd, all_erd = median3(erd_step)
    return dict(value=round(nimg / t_erd, 4), unit="images/sec", cores=physical, kind="port",
                cpu_model=model, logical_cpus=logical,
                sample=f"batch {nimg} at {(H + 31) // 32 * 32}x{(W + 31) // 32 * 32}, median of 3 after 1 warm-up: ERD step (teacher fwd+ERS+NMS+student "
                       f"fwd+losses+backward) {t_erd:.2f} s; oracle/erd_oracle.py on torch-CPU fp32, {physical} threads",
                configs0_gfl_first40_fwd_loss={"value": round(nimg / t_gfl, 4), "unit": "images/sec",
                                               "seconds": round(t_gfl, 3)},
                seconds_all={"erd_step": [round(t, 3) for t in all_erd], "gfl_first40": [round(t, 3) for t in all_gfl]})


def pmc_traffic_per_launch(symbol_prefix: str):
    """HBM-side bytes per launch of the kernels whose symbol starts with `symbol_prefix`, from the committed PMC
    summary (two separate rocprofv3 --pmc passes, tools/pmc_traffic.py; FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process -> None when absent."""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if not files:
        return None, None
    d = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    n = sum(v["launches"] for k, v in d.items() if k.startswith(symbol_prefix) and v.get("traffic_MB"))
    if not n:
        return None, None
    mb = sum(v["launches"] * v["traffic_MB"] for k, v in d.items() if k.startswith(symbol_prefix) and v.get("traffic_MB"))
    return int(mb / n * 1e6), "profiles/" + files[-1]


def pmc_mfma_busy(symbol_prefix: str):
    """matrix-pipe busy fraction of the kernels whose symbol starts with `symbol_prefix` from the committed PMC summary
    (one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass, tools/pmc_mfma.py); None when absent"""
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith("pmc_mfma_busy.json")) if os.path.isdir(pdir) else []
    if not files:
        return None, None
    d = json.load(open(os.path.join(pdir, files[-1])))
    rows = [(v["launches"], v["mfma_busy_fraction"]) for k, v in d.items() if k.startswith(symbol_prefix)]
    n = sum(r[0] for r in rows)
    return (round(sum(a * b for a, b in rows) / n, 4), "profiles/" + files[-1]) if n else (None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (BASELINE configs[1]: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--teacher-graph", action="store_true",
                    help="replay the frozen teacher's pass from a hipGraph (BASELINE configs[4]); same arithmetic")
    ap.add_argument("--compute", choices=["f32", "bf16"], default="f32",
                    help="f32: fp32 matrix cores (BASELINE configs[1], the headline).  bf16: the 1x1/3x3 convolutions on the "
                         "bf16 matrix cores, fp32 accumulation and storage (BASELINE configs[2])")
    ap.add_argument("--step-graph", action="store_true",
                    help="replay the whole step (everything between two SGD updates) from one hipGraph; same arithmetic")
    ap.add_argument("--no-teacher-ahead", action="store_true",
                    help="do not hand the trainer the following batch (teacher of step t+1 then runs next to the forward of step t+1)")
    ap.add_argument("--no-streamk", action="store_true",
                    help="A/B aid: tile-parallel implicit-GEMM launches instead of the stream-K split (every world size uses "
                         "stream-K by default, so the N = 1 point of a scaling curve is the sibling of the N > 1 points)")
    ap.add_argument("--serial", action="store_true",
                    help="no stream concurrency in the timed region either (the rocprofv3 companion run)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("ERD_FORCE_DIST") == "1":      # (the latter: exercise the RCCL path on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # 'nccl' == RCCL on ROCm

    from erd_amd import functional as Fn
    from erd_amd import kernels as K
    from erd_amd.engine import ERDTrainer
    K.set_compute(args.compute)
    if args.no_streamk:
        K.STREAMK = False
    model, cfg = build_model(device, rank)
    opt = cfg.optim_wrapper.optimizer
    trainer = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                         base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=args.batch,
                         auto_scale_lr=cfg.auto_scale_lr.enable, teacher_graph=args.teacher_graph,
                         step_graph=args.step_graph)
    batches = [synthetic_gpu_batch(args.batch, seed=rank * 1000 + i, device=device, cfg=cfg) for i in range(2)]

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
    for j in range(args.warmup):
        log = trainer.train_step(*seq(j), next_batch=nb(j))
    trainer.flush()
    barrier()
    t0 = time.perf_counter()
    for j in range(args.warmup, args.warmup + args.steps):
        log = trainer.train_step(*seq(j), next_batch=nb(j))
    trainer.flush()                   # the deferred SGD of the last step belongs to the timed region
    barrier()
    dt = time.perf_counter() - t0
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
            trainer.train_step(*batches[i % len(batches)])
        trainer.flush()
        ktime = K.timing_end()
        set_serial(args.serial)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    devices = [f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"]
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    dt = float(tmax.item())
    loss = float(log["loss"].detach()) if log is not None else float("nan")

    if rank == 0:
        images = args.batch * world * args.steps
        out = {
            "metric": "images/sec GFL-R50 40+40 incre step @1333x800",
            "value": round(images / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.compute, "data": "synthetic",
            "config": {"workload": "gfl_r50_fpn first_40_incre_last_40 ERD (BASELINE configs[1]), 1333x800 padded to "
                                   "800x1344, " + ("fp32" if args.compute == "f32" else ("bf16 matrix cores, bf16-stored maps, fp32 accumulate / statistics / losses"
                                                                                 if K.BF16_STORAGE else "bf16 multiplicands / fp32 accumulate+storage")) +
                                   ", procedural weights", "batch_per_gpu": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "loss": round(loss, 6), "streams": "serial" if args.serial else ("teacher(t+1)||backward(t), cls||reg towers, trailing weight gradients" if ahead
                                                             else "teacher||student, cls||reg towers, trailing weight gradients"),
            "teacher": "hipGraph replay" if args.teacher_graph else "eager launches",
            "step_graph": bool(trainer.step_graph),
        }
        out["collectives"] = {"backend": "nccl (RCCL)" if dist.is_initialized() else None,
                              "world_size": dist.get_world_size() if dist.is_initialized() else 1, "devices": devices}
        out["kernel_config"] = "stream-K implicit GEMM" if (K.STREAMK and (K.STREAMK_MULTIRANK or not K._multi_rank())) \
            else "tile-parallel implicit GEMM"
        shared = bool(getattr(model, "shares_trunk", lambda: False)()) and not args.teacher_graph
        # student and teacher hold the same frozen stem + layer1: computed once per step and fed to both.  The skipped
        # launches are the student's copy (2.53 + 14.31 GMAC per image, BASELINE.md section 3 / SURVEY Appendix A);
        # `roofline.step_frac` keeps counting the ALGORITHMIC work of the reference's step (both copies)
        out["shared_frozen_trunk"] = {"enabled": shared, "skipped_gflop_per_image": 33.68 if shared else 0.0}
        if ktime:
            dom = max(ktime.values(), key=lambda r: r["ms"])
            peak_tf = FP32_MFMA_PEAK_TFLOPS if args.compute == "f32" else BF16_MFMA_PEAK_TFLOPS   # every GEMM class follows --compute
            sym = {"conv_wgrad": "conv_wgrad", "conv_wino_fwd": "wino", "conv_wino_dgrad": "wino"}.get(dom["kernel"], "conv_igemm")
            # PMC counters cannot be read from inside this process: `traffic` / `mfma_busy_pmc` are STATIC values from the
            # committed profile of the same command (fp32 only; the profile names its commit) -- pointers, not measurements
            traffic, traffic_src = pmc_traffic_per_launch(sym) if args.compute == "f32" else (None, None)
            out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"],
                               "achieved": round(dom["flop"] / (dom["ms"] * 1e-3) / 1e12, 2),
                               "peak": peak_tf, "unit": "TFLOP/s",
                               "frac": round(dom["flop"] / (dom["ms"] * 1e-3) / 1e12 / peak_tf, 4),
                               "traffic": traffic, "traffic_unit": "bytes/launch (L2<->fabric, PMC)",
                               "traffic_source": traffic_src, "traffic_static": True,
                               "pass": f"{rsteps} extra steps, streams serialized",
                               "launches_per_step": dom["launches"] // rsteps,
                               "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                               "gflop_per_launch": round(dom["flop"] / dom["launches"] / 1e9, 3),
                               "algorithmic_bytes_per_launch": int(dom["min_bytes"] / dom["launches"])}
            out["roofline"]["mfma_busy_pmc"], out["roofline"]["mfma_busy_source"] = \
                pmc_mfma_busy(sym) if args.compute == "f32" else (None, None)
            out["roofline"]["mfma_busy_static"] = True
            # step level: the algorithmic work of the WHOLE step (1663.6 GFLOP per image) over the
            # un-instrumented step time of the timed region, against the same peak
            out["roofline"]["step_gflop_per_image"] = STEP_GFLOP_PER_IMAGE
            out["roofline"]["step_tflops"] = round(args.batch * STEP_GFLOP_PER_IMAGE / (1e3 * dt / args.steps), 2)
            out["roofline"]["step_frac"] = round(out["roofline"]["step_tflops"] / peak_tf, 4)
            if dom["kernel"].startswith("conv_wino"):
                # `achieved` counts the direct-convolution flops of the launch (the algorithmic figure); Winograd
                # F(2x2,3x3) executes 16/36 of those multiplications on the matrix cores
                out["roofline"]["executed_flop_fraction"] = round(16 / 36, 4)
                out["roofline"]["mfma_pipe_utilisation"] = round(out["roofline"]["frac"] * 16 / 36, 4)
            out["kernels"] = {k: {"ms_per_step": round(r["ms"] / rsteps, 3),
                                  "tflops": round(r["flop"] / (r["ms"] * 1e-3) / 1e12, 2) if r["flop"] else None,
                                  "launches_per_step": r["launches"] // rsteps} for k, r in ktime.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
