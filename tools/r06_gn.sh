#!/bin/bash
# tools/r06_gn.sh TAG: GroupNorm statistics from the producing convolution: the new test, the GN tests, step A/B (ERD_GN_FUSED=0 / 1)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_gn}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "groupnorm or tower_layer" -p no:cacheprovider 2>&1 | tail -15 | tee $O/tests.txt
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_functions.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee -a $O/tests.txt
for g in 0 1 0 1 0 1; do
  ERD_GN_FUSED=$g python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/b.log 2>&1; echo "ERD_GN_FUSED=$g $(grep '^{' $O/b.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a $O/step_ab.txt
done
