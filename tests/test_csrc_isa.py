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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_lds_dma_kernels_own_m0_and_keep_their_k_loop_free_of_full_waits(tmp_path):
    """Round 6: the three-limb implicit GEMM stages its operand slices by LDS-DMA issued from INLINE ASSEMBLY (`buffer_load_dwordx4 ... offen lds`,
    conv_mfma.hip `glds16`): the statement writes M0 -- a register the compiler reserves but does not preserve around asm -- and its loads are
    absent from the compiler's wait bookkeeping.  Both are safe only under invariants the ISA shows: (i) nothing the COMPILER emitted in
    these kernels reads or writes M0 (so the asm's M0 can neither clobber nor be clobbered), (ii) no scratch memory (a spill reload is a
    vector load: its wait would cover the DMA in flight), (iii) every instantiation carries its DMA statements (ten per K-slice site:
    four activation + six weight-plane loads at 128 x 128, four + three at 128 x 64) and the explicit `s_waitcnt vmcnt(0)` in front of the
    slice barrier."""
    import re
    out = tmp_path / "conv_mfma.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-S", "--cuda-device-only",
                        "-o", str(out), "conv_mfma.hip"], cwd=os.path.join(ROOT, "erd_amd", "csrc"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(out).read()
    seen = 0
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "offen lds" not in body:
            continue
        seen += 1
        assert "conv_igemm_kernel" in name, name
        in_asm, dma, asm_waits = False, 0, 0
        for line in body.split("\n"):
            t = line.strip()
            if "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            elif t and not t.startswith(";"):
                if in_asm:
                    dma += "offen lds" in t
                    asm_waits += t.startswith("s_waitcnt vmcnt(0)")
                    if "m0" in t:
                        assert t.startswith("s_mov_b32 m0") or t.startswith("s_add_i32 m0"), (name, t)
                else:
                    assert not re.search(r"\bm0\b", t), (name, t)              # (i)
                    assert not t.startswith("scratch_"), (name, t)              # (ii)
        assert dma in (14, 20), (name, dma)                                       # (iii): prologue + K-loop site, 128 x 64 / 128 x 128 tiles
        assert asm_waits >= 2, (name, asm_waits)
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0, name
    assert seen == 12, seen        # {128 x 128, 128 x 128 with per-segment taps, 128 x 64} x (residual, mask) presence


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_distillation_nms_kernel_fits_static_lds_and_uses_no_scratch(tmp_path):
    """Round 6: `distill_nms_kernel` keeps up to NMS_LDS_K sorted boxes, their anchors and suppression flags in STATIC LDS (the 64 KB a
    kernel gets without the dynamic-LDS attribute): the kernel descriptor must say so, with no scratch and one workgroup of 1024 threads."""
    import re
    out = tmp_path / "losses.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=off", "-S",
                        "--cuda-device-only", "-o", str(out), "losses.hip"], cwd=os.path.join(ROOT, "erd_amd", "csrc"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    s = open(out).read()
    m = re.search(r"\.amdhsa_kernel \S*distill_nms_kernel\S*\n(.*?)\.end_amdhsa_kernel", s, re.S)
    assert m, "distill_nms_kernel not found"
    desc = m.group(1)
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", desc).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
    src = open(os.path.join(ROOT, "erd_amd", "csrc", "losses.hip")).read()
    cap = int(re.search(r"constexpr int NMS_LDS_K = (\d+);", src).group(1))
    assert cap * (16 + 4 + 4 + 1) <= lds <= 65536, (cap, lds)
    assert scratch == 0
