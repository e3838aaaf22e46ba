"""One rank of tests/test_gpu_dist_world2.py (started by that test, not collected by pytest): the REAL trainer -- ERDTrainer
with every default on -- at world size 2 on ONE GPU.  RCCL refuses two ranks on one device, gloo does not:
ERD_DIST_BACKEND=gloo (erd_amd/dist_utils.py) stages every collective through host memory.  Usage:
    python tests/world2_worker.py RANK WORLD PORT OUTDIR HEIGHT,WIDTH,STEPS,BUCKET_MB,IMAGES_PER_RANK [...]
(several cases run one after the other in the same pair of processes: a rank's start -- interpreter, torch, the HIP runtime, gloo -- is
15-40 s of a test box's time, more than a small case's three steps)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_batches(rank, H, W, bs=2):
    """the two alternating batches of one rank (seeds differ per rank: every rank sees its own images)"""
    from oracle import erd_oracle as O
    out = []
    for s in (0, 1):
        imgs, boxes, labels = O.synthetic_batch(bs, H, W, 40, seed=300 + 10 * rank + s)
        x, metas = O.preprocess(imgs)
        out.append((x, boxes, labels, metas))
    return out


LR, MOM, WD, WARM, WARM_START = 0.01, 0.9, 1e-4, 3, 0.5


def run_case(rank, world, outdir, tag, H, W, steps, bucket_mb, BS):
    import torch
    import torch.distributed as dist
    from erd_amd import dist_utils as DU
    from e2e_util import build_erd, f7_state_dicts, make_samples
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    model = build_erd(tsd, ssd)
    # the reference's optimizer settings through the trainer's own auto-scaling: lr 0.01 x world x bs / 16 (config :112-116)
    tr = ERDTrainer(model, lr=LR, momentum=MOM, weight_decay=WD, base_batch_size=16, batch_size_per_gpu=BS, auto_scale_lr=True,
                    warmup_iters=WARM, warmup_start_factor=WARM_START, bucket_mb=bucket_mb)
    assert tr.distributed and tr.world == world and tr.sync is not None and tr.overlap_teacher and tr.prep is not None
    # record what the loss sends through reduce_mean (the fused C2 + C3 vector) and what comes back
    seen = []
    real = DU.reduce_mean

    def spy(t):
        out = real(t)
        seen.append((t.detach().cpu().clone(), out.detach().cpu().clone()))
        return out
    DU.reduce_mean = spy
    try:
        gpu = [(x.cuda(), make_samples(b, l, m)) for x, b, l, m in rank_batches(rank, H, W, BS)]
        logs, grad0 = [], None
        for it in range(steps):
            lv = tr.train_step(*gpu[it % 2], next_batch=gpu[(it + 1) % 2])
            logs.append({k: float(v) for k, v in lv.items()})
            if it == 0:            # the all-reduced gradient of step 0, before the (deferred) update consumes it
                tr.sync.wait()
                torch.cuda.synchronize()
                grad0 = {n: (p.grad.detach().cpu().clone() / world) for n, p in zip(tr.flat.names, tr.flat.params)}
        tr.flush()
        torch.cuda.synchronize()
    finally:
        DU.reduce_mean = real
    params = {n: p.detach().cpu().clone() for n, p in zip(tr.flat.names, tr.flat.params)}
    torch.save(dict(logs=logs, grad0=grad0, params=params, reduce_mean=seen, late_buckets=tr.sync.late_buckets, missing=list(tr.sync.missing), repeats=sorted(tr.sync.repeats),
                    lrs=[tr.lr_at(i) for i in range(steps)], buckets=len(tr.flat.buckets),
                    backend=dist.get_backend(), device=torch.cuda.current_device()), os.path.join(outdir, f"{tag}_rank{rank}.pt"))
    dist.barrier()
    del tr, model, gpu
    torch.cuda.empty_cache()


def run_reserve_probe(rank, world, outdir):
    """ERDTrainer.tune_cu_reserve with the process group live (the REAL path: real steps at every candidate, timed, one MAX all-reduce):
    both ranks must come out with the same reserve, whatever their own timings said"""
    import torch
    from e2e_util import build_erd, f7_state_dicts, make_samples
    from erd_amd import kernels as K
    from erd_amd.engine import ERDTrainer
    tsd, ssd = f7_state_dicts()
    tr = ERDTrainer(build_erd(tsd, ssd), lr=LR, batch_size_per_gpu=1, auto_scale_lr=False, warmup_iters=0, bucket_mb=4)
    x, b, l, m = rank_batches(rank, 224, 288, 1)[0]
    try:
        res = tr.tune_cu_reserve([(x.cuda(), make_samples(b, l, m))], candidates=(0, 8), steps=1)
        res["library_reserve"] = K.CU_RESERVE
    finally:
        K.set_cu_reserve(0)
    tr.flush()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(outdir, f"reserve_rank{rank}.pt"))


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[5:]]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, ERD_DIST_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from erd_amd import dist_utils as DU
    torch.cuda.set_device(DU.device_index(rank))
    dist.init_process_group(DU.backend_name(), rank=rank, world_size=world)
    try:
        for i, (H, W, steps, bucket_mb, bs) in enumerate(cases):
            run_case(rank, world, outdir, "case%d" % i, H, W, steps, bucket_mb, bs)
        run_reserve_probe(rank, world, outdir)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
