#!/bin/bash
# round 5, GPU call B: guard-band exact predicates (inline) vs the round-4 kernels: pass timings + the tolerance fuzz with NO counted allowance
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_b; mkdir -p $OUT
cd $R
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh shadows,ao,reflections,ddgi r5base base r5base base
done 2>&1 | tee $OUT/ab_guard.txt
export HR_TEST_OUTLIER_PIXELS=0 HR_TEST_DDGI_OUTLIERS=0 HR_TEST_REFL_OUTLIERS=0
( time timeout 1200 python tools/fuzz_tolerance.py 501 60 ) > $OUT/fuzz_strict_new.txt 2>&1; tail -3 $OUT/fuzz_strict_new.txt
( HR_LIBRARY=$R/hybrid_rendering_amd/variants/libhybrid_rendering_amd.r5base.so timeout 1200 python tools/fuzz_tolerance.py 501 60 ) > $OUT/fuzz_strict_base.txt 2>&1; tail -2 $OUT/fuzz_strict_base.txt
grep -c "OUT OF TOLERANCE" $OUT/fuzz_strict_new.txt $OUT/fuzz_strict_base.txt
