#!/bin/bash
# tools/r06_host.sh TAG: host breakdown + multi-stream timeline of both modes on one box (first half)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_host}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
line() { grep '^{' "$1" | tail -1; }
python bench.py --no-cpu-baseline > $O/default_bench.log 2>&1; line $O/default_bench.log > $O/default_bench.json
python bench.py --compute bf16 --no-cpu-baseline > $O/bf16_bench.log 2>&1; line $O/bf16_bench.log > $O/bf16_bench.json
python tools/host_breakdown.py > $O/f32_host_breakdown.txt 2>&1
ERD_COMPUTE=bf16 python tools/host_breakdown.py > $O/bf16_host_breakdown.txt 2>&1
for m in f32 bf16; do
  extra=""; [ $m = bf16 ] && extra="--compute bf16"
  rocprofv3 --kernel-trace -d /tmp/tl_${T}_$m -o tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $O/${m}_tl_bench.log 2>&1
  python tools/timeline.py $(find /tmp/tl_${T}_$m -name "*.db" | head -1) 5 > $O/${m}_timeline.txt 2>&1      # (step 5 of 9: a timed one)
done
tail -3 $O/*_host_breakdown.txt; head -20 $O/*_timeline.txt
python - <<PY
import json
for f in ("default","bf16"):
    d=json.load(open("$O/%s_bench.json"%f)); print(f, d["value"], d["ms_per_step"])
PY
