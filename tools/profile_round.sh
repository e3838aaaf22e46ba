#!/bin/bash
# tools/profile_round.sh TAG: every artefact profiles/ holds for a round, from ONE box (run through gpurun; writes gpurun_out/TAG/).
# rocprofv3 gets the program itself after `--` (python3 ...), never a wrapper; PMC passes are separate runs without tracing domains.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-prof}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
line() { grep '^{' "$1" | tail -1; }
prof() { n=$1; shift; rocprofv3 --kernel-trace --stats -d /tmp/prof_${T}_$n -o $n -- python3 bench.py "$@" > $O/${n}_bench.log 2>&1
         python tools/rocpd_summary.py $(find /tmp/prof_${T}_$n -name "*.db" | head -1) > $O/${n}_kernel_stats.txt 2>>$O/errors.log; line $O/${n}_bench.log > $O/${n}_bench.json; }
pmc() { n=$1; shift; for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${T}_${n}_$c -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2>>$O/errors.log; done
        python tools/pmc_traffic.py /tmp/pmc_${T}_${n}_FETCH_SIZE /tmp/pmc_${T}_${n}_WRITE_SIZE > $O/${n}_pmc_traffic.json 2>>$O/errors.log
        rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_${T}_${n}_mfma -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2>>$O/errors.log
        python tools/pmc_mfma.py /tmp/pmc_${T}_${n}_mfma > $O/${n}_pmc_mfma_busy.json 2>>$O/errors.log; }
# un-profiled lines first (the chip has not been heated by the profiler runs)
python bench.py > $O/default_bench.log 2>&1; line $O/default_bench.log > $O/default_bench.json
python bench.py --compute f32 --no-cpu-baseline > $O/f32native_bench.log 2>&1; line $O/f32native_bench.log > $O/f32native_bench.json
ERD_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/forcedist_bench.log 2>&1; line $O/forcedist_bench.log > $O/forcedist_bench.json
python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 5 > $O/plain20_bench.log 2>&1; line $O/plain20_bench.log > $O/plain20_bench.json
python bench.py --compute bf16 --no-cpu-baseline > $O/bf16_bench.log 2>&1; line $O/bf16_bench.log > $O/bf16_bench.json
python bench.py --arch r101_70_10 --no-cpu-baseline > $O/r101_bench.log 2>&1; line $O/r101_bench.log > $O/r101_bench.json
python bench.py --mixed-res --no-cpu-baseline --steps 20 > $O/mixed_bench.log 2>&1; line $O/mixed_bench.log > $O/mixed_bench.json
python bench.py --mixed-res --compute bf16 --no-cpu-baseline --steps 20 > $O/bf16_mixed_bench.log 2>&1; line $O/bf16_mixed_bench.log > $O/bf16_mixed_bench.json
python tools/step_breakdown.py 3 f32x3 > $O/step_breakdown.txt 2>>$O/errors.log
hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak_$T 2>>$O/errors.log && /tmp/mfma_peak_$T > $O/mfma_peak.txt 2>&1
prof serial --serial --steps 6 --warmup 2 --no-cpu-baseline
prof bf16_serial --serial --steps 6 --warmup 2 --no-cpu-baseline --compute bf16
pmc f32
pmc bf16 --compute bf16
# the default line once more with THIS run's PMC summaries in place (bench.py reads the newest profiles/*_pmc_*.json: the first line above
# was taken before they existed and carries `pmc_stale`)
cp $O/f32_pmc_traffic.json profiles/${T}_f32_pmc_traffic.json; cp $O/f32_pmc_mfma_busy.json profiles/${T}_f32_pmc_mfma_busy.json
cp $O/bf16_pmc_traffic.json profiles/${T}_bf16_pmc_traffic.json; cp $O/bf16_pmc_mfma_busy.json profiles/${T}_bf16_pmc_mfma_busy.json
python bench.py > $O/default_final_bench.log 2>&1; line $O/default_final_bench.log > $O/default_final_bench.json
for f in default default_final f32native forcedist plain20 bf16 r101 mixed bf16_mixed serial bf16_serial; do python - <<PY
import json
try:
    d = json.load(open("$O/${f}_bench.json")); r = d.get("roofline", {})
    print("$f", d["value"], d["ms_per_step"], r.get("kernel"), r.get("frac"), r.get("mfma_executed_frac"), r.get("x_fp32_mfma_ceiling"), r.get("x_fp32_mfma_ceiling_executed"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
