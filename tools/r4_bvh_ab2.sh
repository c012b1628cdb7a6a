#!/bin/bash
# round 4, GPU: reinsertion settings of the BVH build, two repetitions each (the tails of the trace kernels are noisy)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_b; mkdir -p $O
cd $R
export FRAMES=30
{
for rep in 1 2; do
for cfg in "0 0.1" "1 0.02" "1 0.05" "2 0.1" "2 0.3"; do
  set -- $cfg
  echo "#### rep $rep reinsertion passes $1 fraction $2"; HR_BVH_REINSERT=$1 HR_BVH_REINSERT_FRACTION=$2 bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
done
done
} > $O/ab_reinsert_1080p.txt 2>&1
