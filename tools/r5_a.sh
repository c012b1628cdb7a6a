#!/bin/bash
# round 5, GPU call A: the bench line as the driver runs it + the env-switch negatives of rounds 1-2 re-measured on the round-4 tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_a; mkdir -p $OUT
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc $?"; wc -c $OUT/bench_driver.json
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
timeout 900 python -m pytest tests/test_bench_robustness.py tests/test_gpu_shadows.py -m gpu -x -q 2>&1 | tail -5
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2 base"; bash tools/ab.sh shadows,ao,reflections,ddgi base
  echo "#### $1x$2 HR_DDGI_WAVEFRONT=1"; HR_DDGI_WAVEFRONT=1 bash tools/ab.sh ddgi base
  echo "#### $1x$2 HR_TRACE_KERNEL=queue"; HR_TRACE_KERNEL=queue bash tools/ab.sh shadows base
done 2>&1 | tee $OUT/ab_env.txt
