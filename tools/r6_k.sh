#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_k; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_devpaths.py -x -q 2>&1 | tail -30 | tee $OUT/pytest_fullsize.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2 exact gather unroll"; bash tools/ab.sh reflections,ddgi base unroll2 unroll4 base unroll2 2>&1 | sed -E "s/'irradiance_probe.*//; s/'temporal_acc.*//"
done | tee $OUT/ab_exact_gather_unroll.txt
