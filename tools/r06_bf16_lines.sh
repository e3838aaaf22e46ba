#!/bin/bash
# tools/r06_bf16_lines.sh TAG: the bf16-mode lines of tools/profile_round.sh alone (after a host-side change that only touches that mode)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_bf16}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
line() { grep '^{' "$1" | tail -1; }
python bench.py --compute bf16 --no-cpu-baseline > $O/bf16_bench.log 2>&1; line $O/bf16_bench.log > $O/bf16_bench.json
python bench.py --mixed-res --compute bf16 --no-cpu-baseline --steps 20 > $O/bf16_mixed_bench.log 2>&1; line $O/bf16_mixed_bench.log > $O/bf16_mixed_bench.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_${T}_bf16 -o bf16_serial -- python3 bench.py --serial --steps 6 --warmup 2 --no-cpu-baseline --compute bf16 > $O/bf16_serial_bench.log 2>&1
python tools/rocpd_summary.py $(find /tmp/prof_${T}_bf16 -name "*.db" | head -1) > $O/bf16_serial_kernel_stats.txt 2>>$O/errors.log; line $O/bf16_serial_bench.log > $O/bf16_serial_bench.json
python - <<PY
import json
for f in ("bf16", "bf16_mixed", "bf16_serial"):
    d = json.load(open("$O/%s_bench.json" % f)); print(f, d["value"], d["ms_per_step"], {k: (v["launches_per_step"], v["ms_per_step"], v["mfma_frac"], v["hbm_frac"]) for k, v in d["roofline"].get("per_kernel", {}).items()})
PY
