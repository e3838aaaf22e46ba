#!/bin/bash
# same-box sweep of the short-K implicit-GEMM variants over the 1x1 layer shapes (experiment flags of conv_mfma.hip)
SH="L1.conv3 L2.conv1_0 L2.conv1 L2.conv3 L3.conv1 L3.conv3 L4.conv1 L4.conv3 fpn.lat3 L1.conv1_256"
for cfg in "" "ERD_PT=1" "ERD_IG_FORCE=1" "ERD_IG_FORCE=1 ERD_PT=1" "ERD_IG_FORCE=2" "ERD_IG_FORCE=2 ERD_PT=1" "ERD_IG_FORCE=3" "ERD_IG_FORCE=3 ERD_PT=1" "ERD_IG_FORCE=4" "ERD_IG_FORCE=4 ERD_PT=1" "ERD_IG_FORCE=5" "ERD_IG_FORCE=5 ERD_PT=1" "ERD_IG_FORCE=6" "ERD_IG_FORCE=6 ERD_PT=1"; do
  echo "=== $cfg"
  env $cfg python tools/bench_conv.py $SH 2>/dev/null | grep -v amdgpu
done
