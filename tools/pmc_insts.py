#!/usr/bin/env python3
"""Instruction mix of the GEMM-shaped kernels per matrix instruction: SQ_INSTS_* totals from `rocprofv3 --pmc` passes over two
serialized steps (same recipe as tools/pmc_waves.py).  usage: python tools/pmc_insts.py DIR_PASS_A DIR_PASS_B ..."""
import csv, collections, glob, os, sys
KEEP = ("wino_x3_kernel", "conv_igemm_kernel", "conv_thin_x3_kernel", "conv_wgrad_row3_x3_kernel")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            sym = next((s for s in KEEP if k.startswith(s)), None)
            if sym is None:
                continue
            if sym == "conv_wgrad_row3_x3_kernel":
                sym += "<3 taps>" if "true>" in k else "<1 tap>"
            tot[sym][r["Counter_Name"]] = max(tot[sym][r["Counter_Name"]], 0.0) + float(r["Counter_Value"])
names = sorted({n for v in tot.values() for n in v})
base = "SQ_INSTS_VALU_MFMA_MOPS_BF16" if any("SQ_INSTS_VALU_MFMA_MOPS_BF16" in v for v in tot.values()) else "SQ_INSTS_MFMA"
print("instructions per kernel symbol (all launches of two serialized steps), and per matrix instruction (%s)" % "SQ_INSTS_MFMA")
print("%-34s " % "kernel" + " ".join("%18s" % n.replace("SQ_INSTS_", "") for n in names))
for sym, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
    m = c.get("SQ_INSTS_MFMA", 0.0) or 1.0
    print("%-34s " % sym + " ".join("%18s" % ("%.3e" % c.get(n, 0)) for n in names))
    print("%-34s " % "   per MFMA" + " ".join("%18.3f" % (c.get(n, 0) / m) for n in names))
