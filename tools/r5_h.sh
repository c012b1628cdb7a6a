#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_h; mkdir -p $OUT
cd $R
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh shadows,ao r5base base lean6 lean67 noredo r5base base lean6 lean67 noredo
done 2>&1 | tee $OUT/ab_taps2.txt
timeout 300 python -m pytest tests/test_gpu_tile_order.py -x -q 2>&1 | grep -E "^E|assert|passed|failed" | head -20
