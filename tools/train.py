#!/usr/bin/env python3
"""Train a detector from a reference-format config on MI355X (the thin launcher of SURVEY.md 8(f) rank 4).

    python tools/train.py configs/gfl_increment/gfl_r50_fpn_1x_coco_first_40_incre_last_40_cats.py \
        --work-dir work_dirs/erd --cfg-options train_dataloader.batch_size=4 --synthetic 100
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch

Same flags as the reference's tools/train.py:15-129 (--work-dir, --amp, --auto-scale-lr, --resume, --cfg-options,
--launcher).  With the config's COCO annotation file present the real pipeline runs (CocoAnnotations + shuffled,
rank-sharded aspect-ratio batches + the GPU resize/flip/normalise kernel); otherwise, or with `--synthetic ITERS`,
batches shaped like the reference's demo_mm_inputs.  `--amp` selects the bf16 matrix-core mode (`kernels.set_compute("bf16")`:
bf16 multiplicands, fp32 accumulation and storage -- no loss scaling needed, unlike the reference's fp16 autocast).
"""
import argparse
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_cfg_options(items):
    out = {}
    for it in items or []:
        k, v = it.split("=", 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = [s for s in v.split(",")] if "," in v else v
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train a detector (erd_amd)")
    ap.add_argument("config")
    ap.add_argument("--work-dir")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--auto-scale-lr", action="store_true")
    ap.add_argument("--resume", nargs="?", type=str, const="auto")
    ap.add_argument("--cfg-options", nargs="+")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--synthetic", type=int, default=None, metavar="ITERS",
                    help="synthetic iterations per epoch instead of the config's COCO files (the default when they are absent)")
    ap.add_argument("--image-size", type=int, nargs=2, default=(800, 1333), metavar=("H", "W"))
    ap.add_argument("--max-iters", type=int, default=None)
    ap.add_argument("--local_rank", "--local-rank", type=int, default=0)
    args = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from erd_amd import Config
    from erd_amd.runner import Runner, SyntheticDetData

    if args.amp:        # the reference switches its OptimWrapper to AmpOptimWrapper (tools/train.py:82-92)
        from erd_amd import kernels as K
        K.set_compute("bf16")
    cfg = Config.fromfile(args.config)
    cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    cfg.work_dir = args.work_dir or cfg.get("work_dir") or os.path.join(
        "work_dirs", os.path.splitext(os.path.basename(args.config))[0])
    if args.auto_scale_lr:
        if "auto_scale_lr" not in cfg or "base_batch_size" not in cfg.auto_scale_lr:
            raise RuntimeError('Can not find "auto_scale_lr" or "auto_scale_lr.base_batch_size" in your configuration file.')
        cfg.auto_scale_lr.enable = True
    if args.resume == "auto":
        cfg.resume, cfg.load_from = True, None
    elif args.resume is not None:
        cfg.resume, cfg.load_from = True, args.resume

    from erd_amd.dist_utils import backend_name, device_index
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    torch.cuda.set_device(device_index(local_rank))
    if args.launcher == "pytorch":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend_name())      # 'nccl' (= RCCL; default_runtime.py:14) unless ERD_DIST_BACKEND=gloo
    rank = dist.get_rank() if dist.is_initialized() else 0
    head = cfg.model.bbox_head
    ori = cfg.model.get("ori_setting")
    num_new = head.num_classes - (ori.ori_num_classes if ori else 0)
    dcfg = cfg.train_dataloader.dataset
    ann_path = os.path.join(dcfg.get("data_root", ""), dcfg.get("ann_file", ""))
    if args.synthetic is None and os.path.isfile(ann_path):
        from erd_amd.runner import CocoTrainData
        scale = next((t["scale"] for t in dcfg.get("pipeline", []) if t.get("type") == "Resize"), (1333, 800))
        world = dist.get_world_size() if dist.is_initialized() else 1
        data = CocoTrainData(dcfg, int(cfg.train_dataloader.batch_size), scale=tuple(scale), seed=0, rank=rank, world=world,
                             num_workers=int(cfg.train_dataloader.get("num_workers", 0)))
    else:
        data = SyntheticDetData(int(cfg.train_dataloader.batch_size), num_new, args.synthetic or 50, tuple(args.image_size),
                                seed=rank)
    runner = Runner.from_cfg(cfg, data=data)
    runner.train(max_iters=args.max_iters)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
