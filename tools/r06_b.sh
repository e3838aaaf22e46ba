#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r06_b; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dist_smoke.py -x -q -m gpu -k "cu_reserve or tower_layer or groupnorm or bench" -p no:cacheprovider 2>&1 | tail -4 | tee $O/tests.txt
bash tools/ab_env.sh ERD_BUCKET_UPDATE "1 0" 3 --compute bf16 2>&1 | tee $O/bf16_bucket_ab.txt
bash tools/r06_graph.sh r06_graph 2>&1 | tail -12
