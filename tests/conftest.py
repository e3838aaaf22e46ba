import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _native_backtrace_shim():
    """Round 5: one of eleven GPU-suite runs ABORTED in this parent process, in a thread without a Python frame, while it waited for a
    torchrun child.  Round 6 could not reproduce it (tools/dbg/abort_repro.py: 144 children under a context-holding parent, three
    full-suite runs: no abort), so the next occurrence has to name itself: tools/dbg/segv_bt.c prints the NATIVE frames of the
    aborting thread on SIGABRT / SIGSEGV and then hands the signal to faulthandler.  Built into /tmp with gcc, loaded with ctypes;
    any failure here is ignored (the shim is a diagnostic, not a dependency)."""
    try:
        import ctypes
        import subprocess
        src = os.path.join(ROOT, "tools", "dbg", "segv_bt.c")
        so = os.path.join("/tmp", "erd_segv_bt_%d.so" % os.getuid())
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-shared", "-fPIC", "-O1", src, "-o", so], check=True, capture_output=True, timeout=60)
        ctypes.CDLL(so)
    except Exception:
        pass


def pytest_sessionstart(session):
    if session.config.getoption("-m") and "not gpu" not in session.config.getoption("-m"):
        _native_backtrace_shim()


def pytest_collection_modifyitems(config, items):
    """tests/test_gpu_dist_smoke.py first: its tests run bench.py in child processes (torchrun + RCCL) and this process only waits for them --
    better while it holds no GPU context of its own.  (Round 5: one of eleven full-suite runs ABORTED in this parent process, in a thread
    without a Python frame, while it waited for that child after 39 in-process GPU tests; the GPU was fine for the next process.  Cause
    unknown; with nothing initialised here a runtime event of the child's cannot take the test session down with it.)"""
    first = [it for it in items if it.nodeid.startswith("tests/test_gpu_dist_smoke.py") or "/test_gpu_dist_smoke.py" in it.nodeid]
    if first and os.environ.get("ERD_TEST_PLAIN_ORDER", "0") != "1":      # (=1: file order, the order of the round-5 abort -- reproduction runs)
        items[:] = first + [it for it in items if it not in first]


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture()
def oracle_threads():
    """The CPU oracle is the checker of most GPU tests and their time.  On the GPU box's 128-core host torch defaults to 128 threads, where
    an oracle step takes 24 s; on 32 it takes 6 s (bench.py's cpu_baseline sweep: oneDNN oversubscribes).  NOT 16, although that is faster
    still: at 16 threads torch-CPU's backward of the stride-2 bottleneck reference of tests/test_gpu_functions.py is WRONG by 4e-3 on that
    host (tools/dbg/cpu_threads_block_vs_hip.py: forward equal, input gradient different from the 32- / 128-thread and the HIP result) --
    32 is the count every full-size parity test of rounds 3-5 was validated on.  Oracle-heavy test modules opt in (autouse in the module)."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 32))
    yield
    torch.set_num_threads(n)
