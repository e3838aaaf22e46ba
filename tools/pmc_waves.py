#!/usr/bin/env python3
"""What the waves of each GEMM-shaped kernel do with their cycles: per-kernel totals of SQ counters from separate
`rocprofv3 --pmc` passes (tools/profile_round.sh's recipe: `-- python3 bench.py --serial --steps 2 --warmup 1 ...`), as fractions
of SQ_WAVE_CYCLES (wave-resident cycles) / SQ_BUSY_CYCLES.  usage: python tools/pmc_waves.py DIR_PASS_A DIR_PASS_B ... > table"""
import csv, collections, glob, os, sys

KEEP = ("wino_x3_kernel", "conv_igemm_kernel", "conv_thin_x3_kernel", "conv_wgrad_row3_x3_kernel", "gn_apply_kernel", "wgrad_reduce_kernel")
tot = collections.defaultdict(lambda: collections.defaultdict(float))      # fractions of the pass's own SQ_WAVE_CYCLES
launches = collections.defaultdict(int)
base = "SQ_WAVE_CYCLES"
for d in sys.argv[1:]:
    one = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            sym = next((s for s in KEEP if k.startswith(s)), None)
            if sym is None:
                continue
            if sym == "conv_wgrad_row3_x3_kernel":
                sym += "<3 taps>" if "true>" in k else "<1 tap>"
            one[sym][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == base:
                n[sym] += 1
    for sym, c in one.items():          # every pass carries the base counter: normalise inside the pass
        b = c.get(base, 0.0)
        launches[sym] = max(launches[sym], n[sym])
        for name, v in c.items():
            if name != base and b:
                tot[sym][name] = v / b
        tot[sym]["_wave_cycles"] = max(tot[sym]["_wave_cycles"], b)
names = sorted({n for v in tot.values() for n in v if n != "_wave_cycles"})
print("counters per kernel symbol as a fraction of %s of the same pass (all launches of two serialized steps; SQ counters count quad-cycles)" % base)
print("%-34s %8s " % ("kernel", "launches") + " ".join("%22s" % n.replace("SQ_", "") for n in names))
for sym, c in sorted(tot.items(), key=lambda kv: -kv[1]["_wave_cycles"]):
    print("%-34s %8d " % (sym, launches[sym]) + " ".join("%22s" % ("%.3f" % c[n] if n in c else "-") for n in names))
