#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_i; mkdir -p $OUT
cd $R
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh shadows,ao r5base base noredo r5base base noredo
done 2>&1 | tee $OUT/ab_taps3.txt
timeout 300 python -m pytest tests/test_gpu_tile_order.py -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -20
export HR_TEST_OUTLIER_PIXELS=0 HR_TEST_DDGI_OUTLIERS=0 HR_TEST_REFL_OUTLIERS=0
( time timeout 1500 python tools/fuzz_tolerance.py 777 150 ) > $OUT/fuzz_strict_777.txt 2>&1; tail -n 1 $OUT/fuzz_strict_777.txt
grep "OUT OF" $OUT/fuzz_strict_777.txt | cut -c1-300
