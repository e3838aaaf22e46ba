#!/usr/bin/env python3
"""Round 5 saw ONE abort in eleven GPU-suite runs: the pytest parent died ("Fatal Python error: Aborted", no Python frame in the aborting
thread) while it WAITED for the torchrun child of tests/test_gpu_dist_smoke.py after 39 in-process GPU tests.  This script is that
situation in a loop: a parent that holds a live HIP context (model built, a few training steps run, streams and workspaces allocated)
starts the same child N times and only waits.  Run it under the SIGABRT backtrace shim:
    gcc -shared -fPIC -O1 tools/dbg/segv_bt.c -o /tmp/segv_bt.so && LD_PRELOAD=/tmp/segv_bt.so PYTHONFAULTHANDLER=1 python tools/dbg/abort_repro.py 15
"""
import faulthandler, json, os, subprocess, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from erd_amd.engine import ERDTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hold = (sys.argv[2] if len(sys.argv) > 2 else "ctx") == "ctx"
if hold:
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model(dev, 0)
    opt = cfg.optim_wrapper.optimizer
    tr = ERDTrainer(model, lr=opt.lr, momentum=opt.momentum, weight_decay=opt.weight_decay,
                    base_batch_size=cfg.auto_scale_lr.base_batch_size, batch_size_per_gpu=2, auto_scale_lr=cfg.auto_scale_lr.enable)
    b = bench.synthetic_gpu_batch(2, seed=0, device=dev, cfg=cfg)
    for _ in range(3):
        tr.train_step(*b)
    tr.flush(); torch.cuda.synchronize()
    print("parent holds a HIP context: %.1f GB allocated" % (torch.cuda.memory_allocated() / 2**30), flush=True)
env = dict(os.environ, ERD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
env.pop("LD_PRELOAD", None)
common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-strict-fp32"]
ok = 0
for i in range(n):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + i % 50), os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=600)
    good = r.returncode == 0 and any(l.startswith("{") for l in r.stdout.splitlines())
    ok += good
    print("child %d: rc %d, %.1f s%s" % (i, r.returncode, time.time() - t0, "" if good else "  STDERR: " + r.stderr[-1500:]), flush=True)
    if hold:
        tr.train_step(*b); tr.flush(); torch.cuda.synchronize()      # the parent's context is still alive and usable
print("%d of %d children fine; the parent survived" % (ok, n))
