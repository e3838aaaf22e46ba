export TMPDIR=/tmp
O=gpurun_out/r4e_wino_probes.txt
: > $O
for lib in "" wx3rd8 wx3nosplit wx3nomfma wx3nomfma_nosplit wx3noload wx3prio0; do
  echo "== lib ${lib:-production}" >> $O
  if [ -n "$lib" ]; then export ERD_HIP_LIB=$PWD/erd_amd/lib/abl/liberd_hip_$lib.so; else unset ERD_HIP_LIB; fi
  ONLY="head tower 5 levels,L3.conv2,fpn.out P3" timeout 200 python tools/bench_wino.py 2>/dev/null | grep -v "^layer" >> $O
done
unset ERD_HIP_LIB
cat $O
