#!/bin/bash
# tools/r06_graph.sh TAG: whole-step hipGraph replay against eager launches, both modes: un-profiled rates first, then one kernel-trace timeline each
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_graph}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
val() { grep '^{' "$1" | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
for m in f32x3 bf16; do for g in "" "--step-graph"; do
  python bench.py --compute $m $g --no-cpu-baseline --no-kernel-timing --no-strict-fp32 --steps 20 --warmup 5 > $O/b.log 2>&1
  echo "$m ${g:-eager}: $(val $O/b.log)" | tee -a $O/rates.txt
done; done
for m in f32x3 bf16; do for g in eager graph; do
  extra=""; [ $g = graph ] && extra="--step-graph"
  rocprofv3 --kernel-trace -d /tmp/tl_${T}_${m}_$g -o tl -- python3 bench.py --compute $m $extra --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-strict-fp32 > $O/${m}_${g}_tl.log 2>&1
  python tools/timeline.py $(find /tmp/tl_${T}_${m}_$g -name "*.db" | head -1) 5 > $O/${m}_${g}_timeline.txt 2>&1
  head -1 $O/${m}_${g}_timeline.txt
done; done
