#!/bin/bash
# the thin-K kernel's forms side by side on one box (build: tools/build_probe.sh NAME conv_thin.hip FLAGS):
#   shipped library | thinsl: this tree's conv_thin.hip | thindirect: -DERD_THIN_DIRECT | *_tr: + -DERD_THIN_TRACE (phase trace)
cd "$(dirname "$0")/../.."
run() { echo "== $1"; shift; env "$@" timeout 300 python tools/dbg/thin_trace.py 2>&1 | grep -v amdgpu.ids; }
A=erd_amd/lib/abl/liberd_hip
run "shipped library" X=1
run "thinsl (straight-line block body)" ERD_HIP_LIB=${A}_thinsl.so
run "thinsl_tr" ERD_HIP_LIB=${A}_thinsl_tr.so
run "thindirect" ERD_HIP_LIB=${A}_thindirect.so
run "thindirect_tr" ERD_HIP_LIB=${A}_thindirect_tr.so
run "thinsl, 2 workgroups per CU" ERD_HIP_LIB=${A}_thinsl.so ERD_THIN_WGS=2
run "thinsl, 3 workgroups per CU" ERD_HIP_LIB=${A}_thinsl.so ERD_THIN_WGS=3
run "shipped library again" X=1
