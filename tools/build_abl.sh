#!/bin/bash
# tools/build_abl.sh NAME [extra hipcc flags...]: a liberd_hip variant with the Winograd cycle trace compiled in
# (erd_amd/lib/abl/liberd_hip_NAME.so; read back with tools/dbg/trace.py through ERD_HIP_LIB)
cd "$(dirname "$0")/../erd_amd/csrc"
mkdir -p ../lib/abl
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -DERD_WINO_TRACE "$@" -c winograd.hip -o /tmp/winograd_abl_$n.o 2>/dev/null &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DERD_CSRC_SHA=\"probe:$n\" -c elementwise.hip -o /tmp/elementwise_abl_$n.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_mfma.o conv_thin.o /tmp/elementwise_abl_$n.o losses.o predict.o leaf_ops.o prep.o /tmp/winograd_abl_$n.o -o ../lib/abl/liberd_hip_$n.so
