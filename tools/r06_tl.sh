#!/bin/bash
# tools/r06_tl.sh TAG: multi-stream timeline of one steady step, both modes (rocprofv3 --kernel-trace of the default bench run)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_tl}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
for m in f32 bf16; do
  extra=""; [ $m = bf16 ] && extra="--compute bf16"
  rocprofv3 --kernel-trace -d /tmp/tl_${T}_$m -o tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $O/${m}_tl_bench.log 2>&1
  python tools/timeline.py $(find /tmp/tl_${T}_$m -name "*.db" | head -1) 5 > $O/${m}_timeline.txt 2>&1
done
head -50 $O/f32_timeline.txt; head -50 $O/bf16_timeline.txt
