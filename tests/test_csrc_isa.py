"""CPU: what the compiler makes of the thin-K kernel's block body (erd_amd/csrc/conv_thin.hip, hipcc cross-compiles without a GPU).
Round 5's find (EXPERIMENTS 7f): gfx950 retires vector loads AND stores through one in-order counter; with guarded stores and loads
between them the compiler put `s_waitcnt vmcnt(0)` in front of every store and each store waited for the previous one's acknowledgement.
The invariants the rewrite established are properties of the ISA, so they are checked on the ISA: every instantiation issues its four row
stores with no full wait between them, touches no scratch memory (a run-time-indexed register array becomes scratch, whose loads count in
the same counter) and contains no packed fp32 arithmetic (-fno-slp-vectorize: ~13 extra cycles each beside an MFMA stream)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_thin_kernel_stores_back_to_back_without_scratch_or_packed_fp32(tmp_path):
    from vmcnt_audit import audit
    mk = open(os.path.join(ROOT, "erd_amd", "csrc", "Makefile")).read()
    assert "GEMM_FLAGS := -fno-slp-vectorize" in mk and "conv_thin.o" in mk          # the flags below are the Makefile's for this file
    out = tmp_path / "conv_thin.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fno-slp-vectorize", "-S",
                        "--cuda-device-only", "-o", str(out), "conv_thin.hip"], cwd=os.path.join(ROOT, "erd_amd", "csrc"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = {n: v for n, v in audit(str(out)).items() if "conv_thin_x3_kernel" in n}
    assert len(kernels) == 8, sorted(kernels)              # K in {64, 128} x (residual, mask) presence
    for n, v in kernels.items():
        assert v["stores"] == 4 and v["serialized"] == 0, (n, v)
        assert v["scratch"] == 0 and v["packed_f32"] == 0, (n, v)
