#!/bin/bash
# round 4, GPU: the temporal kernels reprojecting from the pass's own geometry records (HR_GEO_HISTORY=1, default) against the caller's previous G-buffer (=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_c; mkdir -p $O
cd $R
export FRAMES=40
{
for rep in 1 2; do
for g in 0 1; do
  echo "#### rep $rep 1920x1080 HR_GEO_HISTORY=$g"; HR_GEO_HISTORY=$g bash tools/ab.sh ${PASSES:-shadows} base | grep -v "^=="
  echo "#### rep $rep 3840x2160 HR_GEO_HISTORY=$g"; PB_ARGS="--width 3840 --height 2160" HR_GEO_HISTORY=$g bash tools/ab.sh ${PASSES:-shadows} base | grep -v "^=="
done
done
} > $O/ab_geo.txt 2>&1
