cd $GRAFT_REPO_ROOT
for sh in 0 1; do ERD_WINO_SHAPES=$sh timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_s$sh.json; done
