#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r06_c; mkdir -p $O
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "hipgraph or step_graph" -p no:cacheprovider 2>&1 | tail -4 | tee $O/tests.txt
val() { grep '^{' "$1" | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; }
for m in f32x3 bf16; do for g in "" "--step-graph" "" "--step-graph"; do
  python bench.py --compute $m $g --no-cpu-baseline --no-kernel-timing --no-strict-fp32 --steps 20 --warmup 5 > $O/b.log 2>&1
  echo "$m ${g:-eager}: $(val $O/b.log)" | tee -a $O/rates.txt
done; done
bash tools/ab_env.sh ERD_BUCKET_UPDATE "1 0" 3 2>&1 | tee $O/f32_bucket_ab.txt
