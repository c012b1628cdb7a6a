#!/bin/bash
# round 5, GPU call R: trace_any_share (idle lanes take pending subtrees of the last live lanes) in the shadow trace — parity, then timing
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_r; mkdir -p $OUT
cd $R
HR_LIBRARY=$R/hybrid_rendering_amd/variants/libhybrid_rendering_amd.share.so timeout 900 python -m pytest tests/test_gpu_shadows.py tests/test_gpu_fullsize.py tests/test_gpu_tile_order.py tests/test_gpu_edge.py -q -x 2>&1 | tail -4
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  echo "#### $1x$2"; bash tools/ab.sh shadows base share share32 share8 base share share32 share8 2>&1 | sed -E "s/'temporal.*//"
done | tee $OUT/ab_share.txt
export PB_ARGS="--width 1920 --height 1080 --tier hard"; echo "#### hard tier"; bash tools/ab.sh shadows base share base share 2>&1 | sed -E "s/'temporal.*//" | tee -a $OUT/ab_share.txt
