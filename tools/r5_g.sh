#!/bin/bash
# round 5, GPU call G: exact tap verdicts (note + cold redo) in the shadow / AO temporal kernels, DDGI redo restricted to the overflow regime
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_g; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh shadows,ao,ddgi r5base base eu6 r5base base eu6
done 2>&1 | tee $OUT/ab_taps.txt
export HR_TEST_OUTLIER_PIXELS=0 HR_TEST_DDGI_OUTLIERS=0 HR_TEST_REFL_OUTLIERS=0
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_strict_new.txt 2>&1; tail -n 1 $OUT/fuzz_strict_new.txt
grep "OUT OF" $OUT/fuzz_strict_new.txt | cut -c1-300
