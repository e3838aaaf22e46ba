#!/usr/bin/env python3
"""Matrix-pipe busy fraction per kernel from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma \\
        -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    python tools/pmc_mfma.py gpurun_out/pmc_mfma > profiles/rNN_pmc_mfma_busy.json

busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs, so the
active cycles of a launch are GRBM_GUI_ACTIVE / 8 (check: the three-tap weight gradient reads 0.80 here and runs at
119-131 of 157.3 TFLOP/s with 7.7 % chunk padding; the Winograd kernel reads 0.52 = its 0.44 algorithmic pipe share plus
15 % tile padding)."""
import collections, csv, glob, json, os, re, sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    a, b = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if a > 0 and b > 0:
        out[k] = dict(launches=cnt[k], mfma_busy_fraction=round(b / (a / 8 * 1024), 4),
                      gui_active_cycles_per_launch=round(a / 8 / cnt[k], 1))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erd_amd import _lib
out["_meta"] = {"csrc_sha256": _lib.load().erd_csrc_sha().decode()}     # the library the counters were taken on
json.dump(out, sys.stdout, indent=1)
