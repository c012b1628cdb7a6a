#!/bin/bash
# round 4, GPU: reinsertion restricted to nodes below a share of the root's area
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_b; mkdir -p $O
cd $R
export FRAMES=30
{
for rep in 1 2; do
for ma in 1.0 0.05 0.01; do
  echo "#### rep $rep standard 1080p, max area $ma"; HR_BVH_REINSERT_MAX_AREA=$ma bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
done
echo "#### rep $rep standard 1080p, no reinsertion"; HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
done
for ma in 1.0 0.05 0.01; do
  echo "#### hard 1080p, max area $ma"; PB_ARGS="--tier hard" HR_BVH_REINSERT_MAX_AREA=$ma bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
done
} > $O/ab_maxarea.txt 2>&1
