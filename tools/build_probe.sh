#!/bin/bash
# tools/build_probe.sh NAME FILE.hip [extra hipcc flags...]: a liberd_hip variant with FILE.hip rebuilt under extra flags
# (erd_amd/lib/abl/liberd_hip_NAME.so; select it with ERD_HIP_LIB for same-box A/B timing runs)
set -e
cd "$(dirname "$0")/../erd_amd/csrc"
mkdir -p ../lib/abl
n=$1; f=$2; shift 2
base=${PROBE_BASE:-$(basename $f .hip)}
extra=""
case $base in losses|leaf_ops|predict) extra="-ffp-contract=off";; conv_thin|winograd) extra="-fno-slp-vectorize";; esac      # (the Makefile's per-file flags)
sha=""; [ $base = elementwise ] && sha="-DERD_CSRC_SHA=\"probe:$n\""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra $sha "$@" -c $f -o /tmp/${base}_probe_$n.o
# the variant reports its own source sha ("probe:NAME"): a PMC summary taken on it cannot pass for the shipped library's (bench.py pmc_stale)
if [ $base != elementwise ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DERD_CSRC_SHA=\"probe:$n\" -c elementwise.hip -o /tmp/elementwise_probe_$n.o
fi
objs=""
for o in conv_mfma conv_thin elementwise losses predict winograd leaf_ops prep; do
  if [ $o = $base ] || [ $o = elementwise ]; then objs="$objs /tmp/${o}_probe_$n.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/abl/liberd_hip_$n.so
echo built erd_amd/lib/abl/liberd_hip_$n.so
