"""bench.py's launcher half runs without a GPU: `--gpus N` outside a torchrun environment starts its own N ranks as child
processes (tools/dist_train.sh:11-19 in the reference) and relays their exit status; the parent itself never initialises
the GPU (it would not be allowed to start other GPU programs afterwards on this pool)."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(['{"metric": "x", "value": 1.0}\n'])

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    args = types.SimpleNamespace(gpus=8)
    rc = bench.self_launch(args, ["--gpus", "8", "--steps", "5", "--launcher", "spawn", "--warmup", "2"])
    assert rc == 7                                                    # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--launcher", "none"]     # same flags; the children never launch
    # the switch is removed by POSITION: `--launcher auto`, `--launcher=spawn` and a value that merely equals "spawn" elsewhere
    for argv, want in ((["--gpus", "8", "--launcher", "auto"], ["--gpus", "8"]),
                       (["--launcher=spawn", "--gpus", "8"], ["--gpus", "8"]),
                       (["--gpus", "8", "--arch", "spawn"], ["--gpus", "8", "--arch", "spawn"])):
        bench.self_launch(args, argv)
        cmd = seen["cmd"]
        assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == want + ["--launcher", "none"], (argv, cmd)
    assert seen["env"]["ERD_BENCH_CHILD"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_2_on_a_cpu_box_fails_in_the_children_not_in_the_parent():
    """no GPU here: the two ranks the parent starts refuse ("needs an MI355X") and the parent exits non-zero -- the parent
    got as far as starting them, i.e. it did not stop at the old 'launch with torch.distributed.run' refusal"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "launch with torch.distributed.run" not in r.stderr.split("needs an MI355X")[0]


def test_committed_pmc_summaries_describe_the_kernel_sources_at_head():
    """bench.py's `roofline.traffic` / `mfma_busy_pmc` / `per_kernel.pmc_*` are static reads of the newest profiles/*_pmc_*.json
    (chosen by file name): each summary records the sha256 of the kernel sources of the library it was taken on, and the newest one of
    each kind must be the sha of erd_amd/csrc/ as committed -- a kernel change without a fresh `tools/profile_round.sh` fails here
    instead of shipping counters of other code; the library carries the same sha (erd_csrc_sha())."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    from csrc_sha import csrc_sha256
    from erd_amd import _lib
    head = csrc_sha256(ROOT)
    assert _lib.load().erd_csrc_sha().decode() == head, "liberd_hip.so was not built from the sources in the tree: make -C erd_amd/csrc"
    for compute in ("f32", "bf16"):
        for suffix in ("pmc_traffic.json", "pmc_mfma_busy.json"):
            f = bench._pmc_file(suffix, compute)
            assert f is not None, (compute, suffix)
            got = json.load(open(f)).get("_meta", {}).get("csrc_sha256")
            assert got == head, f"{os.path.relpath(f, ROOT)} was taken on other kernel sources ({got}): rerun tools/profile_round.sh"
    prov = bench.pmc_provenance("f32")
    assert prov["pmc_stale"] is False and prov["library_csrc_sha"] == head
