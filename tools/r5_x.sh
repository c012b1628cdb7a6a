#!/bin/bash
# round 5, GPU call X: AO entry-node grid — parity, then A/B against the per-pixel descent (HR_AO_ENTRY_GRID=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_x; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ao.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py tests/test_gpu_tile_order.py tests/test_gpu_tiling.py tests/test_gpu_configs4.py -q -x 2>&1 | tail -3
for res in "1920 1080" "3840 2160"; do
  set -- $res
  export PB_ARGS="--width $1 --height $2"
  for g in 1 0 1 0; do echo "#### $1x$2 HR_AO_ENTRY_GRID=$g"; HR_AO_ENTRY_GRID=$g bash tools/ab.sh ao base 2>&1 | grep ao_; done
done | tee $OUT/ab_ao_grid.txt
export PB_ARGS="--width 1920 --height 1080 --tier hard"
for g in 1 0; do echo "#### hard tier HR_AO_ENTRY_GRID=$g"; HR_AO_ENTRY_GRID=$g bash tools/ab.sh ao base 2>&1 | grep ao_; done | tee -a $OUT/ab_ao_grid.txt
