#!/bin/bash
# round 6, GPU call G: cooperative parity gather — threshold A/B (points per wave up to which the shared form is used)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_g; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_reflections.py tests/test_gpu_ddgi.py tests/test_gpu_ref_shaders.py -x -q 2>&1 | tail -3
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh reflections,ddgi coop0 coop24 coop40 base coop56 coop64 coop0 base 2>&1 | grep -v "^ *ddgi.*sample" 
done | tee $OUT/ab_coop_threshold.txt
