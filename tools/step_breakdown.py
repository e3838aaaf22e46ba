#!/usr/bin/env python3
"""Per-layer-shape time table of the GEMM-shaped launches of one serialized ERD step (HIP events per launch):
which shapes own the step, at what (algorithmic) TFLOP/s, against the bound of the pipe the launch actually runs on:
    bound = max(EXECUTED flop / peak of that pipe, algorithmic bytes / 8 TB/s)
  executed flop = algorithmic x 16/36 for Winograd F(2x2,3x3) launches (fp32 MFMA, 157.3 TF), x 6 for three-limb launches
  ("f32x3": six bf16 limb products per fp32 product, bf16 MFMA 2 500 TF), x 1 otherwise (fp32 MFMA; bf16 mode: 2 500 TF).
usage: python tools/step_breakdown.py [steps] [f32x3|f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from erd_amd import functional as Fn
from erd_amd import kernels as K
from erd_amd.engine import ERDTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else K.DEFAULT_COMPUTE
K.set_compute(mode)
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev, 0)
opt = cfg.optim_wrapper.optimizer
tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=4, auto_scale_lr=cfg.auto_scale_lr.enable)
batches = [bench.synthetic_gpu_batch(4, seed=i, device=dev, cfg=cfg) for i in range(2)]
tr.overlap_teacher = False
Fn.TOWERS_ON_TWO_STREAMS = False
Fn.WGRAD_TRAIL = False          # (trailing weight gradients would overlap the input-gradient launches being timed)
for i in range(2):
    tr.train_step(*batches[i % 2])
tr.flush(); torch.cuda.synchronize()
K.TIMING_DETAIL = True
K.timing_begin()
for i in range(steps):
    tr.train_step(*batches[i % 2])
tr.flush()
rec = K.timing_end()
rows = sorted(rec.values(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows) / steps
X3_CLASSES = ("conv_igemm_fwd", "conv_igemm_dgrad", "conv_thin_fwd", "conv_thin_dgrad", "conv_wgrad_row3", "conv_wgrad")


def pipe_of(kernel: str):
    """(executed flop per algorithmic flop, peak of the pipe in flop/s, label) of a timing class"""
    cls = kernel.split(" ")[0]
    if cls.startswith("conv_wino"):
        if K.wino_x3():       # three-limb Winograd: 16/36 of the direct count, six bf16 limb products each
            return bench.WINO_EXECUTED * 6.0, bench.BF16_MFMA_PEAK_TFLOPS * 1e12, "wx3"
        return bench.WINO_EXECUTED, bench.FP32_MFMA_PEAK_TFLOPS * 1e12, "f32"
    if mode == "bf16":
        return 1.0, bench.BF16_MFMA_PEAK_TFLOPS * 1e12, "bf16"
    if mode == "f32x3" and cls in X3_CLASSES:
        return 6.0, bench.BF16_MFMA_PEAK_TFLOPS * 1e12, "x3"
    return 1.0, bench.FP32_MFMA_PEAK_TFLOPS * 1e12, "f32"


print(f"compute mode {mode}; bound = max(executed flop / pipe peak, algorithmic bytes / 8 TB/s)")
print(f"{'kernel / shape':62s} {'n/step':>6s} {'ms/step':>8s} {'us/launch':>9s} {'alg TF':>6s} {'pipe':>4s} {'mfma us':>7s} {'hbm us':>6s} {'x bound':>7s}")
tot_bound = 0.0
for r in rows:
    n = r["launches"] / steps
    us = 1e3 * r["ms"] / r["launches"]
    fl, by = r["flop"] / r["launches"], r["min_bytes"] / r["launches"]
    ex, peak, label = pipe_of(r["kernel"])
    b_mfma, b_hbm = fl * ex / peak * 1e6, by / (bench.HBM_PEAK_GBS * 1e9) * 1e6
    bound = max(b_mfma, b_hbm)
    tot_bound += bound * n
    print(f"{r['kernel']:62s} {n:6.1f} {r['ms'] / steps:8.3f} {us:9.1f} {fl / us / 1e6:6.1f} {label:>4s} {b_mfma:7.1f} {b_hbm:6.1f} {us / bound:7.2f}")
print(f"total {tot:.2f} ms/step over {len(rows)} shapes; sum of the per-launch bounds {tot_bound / 1e3:.2f} ms/step")
