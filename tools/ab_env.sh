#!/bin/bash
# tools/ab_env.sh VAR "v0 v1" [pairs] [extra bench args]: alternating 20-step bench runs with VAR=v0 / VAR=v1 on one box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
VAR=$1; VALS=$2; PAIRS=${3:-3}; shift 3 2>/dev/null || shift $#
for i in $(seq $PAIRS); do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 "$@" > /tmp/ab.log 2>&1
  echo "$VAR=$v $(grep '^{' /tmp/ab.log | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done
