#!/bin/bash
# tools/r06_glds.sh TAG: LDS-DMA operand staging of the three-limb implicit GEMM against the register-staged form (ERD_IG_GLDS=0): bits + time
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-r06_glds}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
ERD_IG_GLDS=0 python tools/dbg/epi_bitcompare.py /tmp/ref.pt > $O/bits_reg.txt 2>&1
ERD_IG_GLDS=1 python tools/dbg/epi_bitcompare.py /tmp/new.pt /tmp/ref.pt > $O/bits_glds.txt 2>&1
tail -3 $O/bits_glds.txt
paste <(grep ' us$' $O/bits_reg.txt) <(grep ' us$' $O/bits_glds.txt | awk '{print $(NF-1)}') | grep igemm
for g in 0 1 0 1; do
  ERD_IG_GLDS=$g python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/b.log 2>&1; echo "ERD_IG_GLDS=$g $(grep '^{' $O/b.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a $O/step_ab.txt
done
