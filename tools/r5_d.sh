#!/bin/bash
# round 5, GPU call D: DDGI gather redo (leak test + Chebyshev note; variant nocheb = leak test only) vs round 4: suite, timings, strict fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_d; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh reflections,ddgi r5base base nocheb r5base base nocheb
done 2>&1 | tee $OUT/ab_redo.txt
export HR_TEST_OUTLIER_PIXELS=0 HR_TEST_DDGI_OUTLIERS=0 HR_TEST_REFL_OUTLIERS=0
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_strict_new.txt 2>&1; tail -1 $OUT/fuzz_strict_new.txt
( HR_LIBRARY=$R/hybrid_rendering_amd/variants/libhybrid_rendering_amd.nocheb.so timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_strict_nocheb.txt 2>&1; tail -1 $OUT/fuzz_strict_nocheb.txt
grep "OUT OF" $OUT/fuzz_strict_new.txt | cut -c1-300
