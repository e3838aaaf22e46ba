#!/bin/bash
# tools/build_abl.sh N...: liberd_hip variants with compile-time ablations of the Winograd kernel (erd_amd/lib/abl/liberd_hip_N.so)
cd "$(dirname "$0")/../erd_amd/csrc"
mkdir -p ../lib/abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DERD_WINO_ABL=$n $EXTRA -c winograd.hip -o /tmp/winograd_abl_$n.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o elementwise.o losses.o predict.o /tmp/winograd_abl_$n.o -o ../lib/abl/liberd_hip_$n.so &
done
wait
