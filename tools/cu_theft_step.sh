#!/bin/bash
# tools/cu_theft_step.sh [OUT]: the default step with k CUs held by a spin kernel for whole steps (bench.py --occupy-cus k), k = 0 / 4 / 8 / 16,
# with the one-dispatch-round split of the three-limb weight gradients (768 workgroups, the default) and with a two-round split (1536).
# A single-GPU proxy for RCCL channels resident beside the backward pass.  Run through gpurun.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=${1:-gpurun_out/cu_theft.txt}
# (the trainer uses five HIP streams; with ROCm's default of four hardware queues the spin kernel shares a queue with one of them and the
#  step simply WAITS for it -- 1.29 s for the first step, tools/dbg/occupy_probe.py.  Eight queues: everything runs side by side.)
export GPU_MAX_HW_QUEUES=8
echo "# GPU_MAX_HW_QUEUES=8 bench.py --occupy-cus k --no-cpu-baseline --no-kernel-timing --no-strict-fp32 --steps 16 --warmup 4 (bs 4, f32x3), one box; img/s" > $O
for tgt in 768 1536; do
  for k in 0 4 8 16; do
    v=$(ERD_WGRAD_ROW3_X3_TARGET=$tgt ERD_WGRAD_X3_TARGET=$tgt timeout 300 python bench.py --occupy-cus $k --no-cpu-baseline --no-kernel-timing --no-strict-fp32 --steps 16 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o = d.get('occupied_cus', {})
print(d['value'], d['ms_per_step'], o.get('held_for_the_whole_timed_region'))")
    echo "weight-gradient split target $tgt  k=$k CUs held: $v" >> $O
  done
done
cat $O
