#!/bin/bash
# tools/r06_a.sh TAG: stream-K threshold A/B on the K = 256 shapes + the multi-stream timeline of both modes
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_a}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
line() { grep '^{' "$1" | tail -1; }
for k in 512 256 128; do
  echo "== ERD_SK_MIN_K=$k" >> $O/sk_min_k.txt
  ERD_SK_MIN_K=$k python tools/bench_conv.py L1.conv1_256 L2.conv1_0 L2.down L3.conv3 L3.conv1 L4.conv3 fpn.lat3 >> $O/sk_min_k.txt 2>&1
done
for k in 512 256 512 256; do
  ERD_SK_MIN_K=$k python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/b.log 2>&1; echo "ERD_SK_MIN_K=$k $(line $O/b.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/sk_min_k.txt
done
for m in f32 bf16; do
  extra=""; [ $m = bf16 ] && extra="--compute bf16"
  rocprofv3 --kernel-trace -d /tmp/tl_${T}_$m -o tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $O/${m}_tl_bench.log 2>&1
  python tools/timeline.py $(find /tmp/tl_${T}_$m -name "*.db" | head -1) > $O/${m}_timeline.txt 2>&1
done
cat $O/sk_min_k.txt; head -40 $O/f32_timeline.txt
