#!/bin/bash
# round 5, GPU call C: Chebyshev knife-edge redo (ddgi_sample_fast.h) vs the round-4 kernels: timings, the strict tolerance fuzz, the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_c; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh reflections,ddgi r5base base r5base base
done 2>&1 | tee $OUT/ab_cheb.txt
export HR_TEST_OUTLIER_PIXELS=0 HR_TEST_DDGI_OUTLIERS=0 HR_TEST_REFL_OUTLIERS=0
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_strict_new.txt 2>&1; tail -3 $OUT/fuzz_strict_new.txt
grep "OUT OF" $OUT/fuzz_strict_new.txt | cut -c1-330
