"""GPU: the RCCL data-parallel path end to end on ONE GPU (world_size 1 under torch.distributed.run):
post-accumulate hooks -> per-bucket async all-reduce on the flat gradient buffer -> deferred fused SGD, and
the fused 2-float all-reduce of the loss normalisers.  (World sizes > 1 are covered by the gloo CPU test and
by the driver's multi-GPU run.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_world1_matches_plain_run():
    env = dict(os.environ, ERD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-strict-fp32"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py")] + common,
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True,
                       env=dict(os.environ), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    e = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["loss"] == pytest.approx(e["loss"], rel=1e-5)      # same arithmetic with and without the collectives


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a torchrun environment starts its own N ranks as child processes (the reference's
    tools/dist_train.sh:11-19 does the launching too); `--launcher spawn` takes that path at N = 1.  The parent relays rank 0's
    JSON line, which says who launched the ranks and that the RCCL process group was up."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "spawn", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-strict-fp32"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["collectives"]["backend"] == "nccl (RCCL)" and d["collectives"]["world_size"] == 1
    assert d["collectives"]["launched_by"].startswith("bench.py self_launch")
    # a failing child is a failing parent (exit status relayed)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "spawn", "--steps", "1", "--warmup", "0",
                          "--batch", "0", "--no-cpu-baseline", "--no-kernel-timing", "--no-strict-fp32"], capture_output=True, text=True, env=env, timeout=600)
    assert bad.returncode != 0
