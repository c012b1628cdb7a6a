#!/bin/bash
# round 4, GPU: reinsertion on / off on the hard tier (1080p) and at 4K (standard tier)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_b; mkdir -p $O
cd $R
export FRAMES=20
{
for cfg in "0 0.1" "2 0.1" "2 0.3"; do
  set -- $cfg
  echo "#### hard tier 1080p, reinsertion passes $1 fraction $2"; PB_ARGS="--tier hard" HR_BVH_STATS=1 HR_BVH_REINSERT=$1 HR_BVH_REINSERT_FRACTION=$2 bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
  echo "#### standard tier 3840x2160, reinsertion passes $1 fraction $2"; PB_ARGS="--width 3840 --height 2160" HR_BVH_REINSERT=$1 HR_BVH_REINSERT_FRACTION=$2 bash tools/ab.sh shadows,ao,reflections,ddgi base | grep -v "^=="
done
echo "#### hard tier 1080p, round-3 tree + slot order"; PB_ARGS="--tier hard" HR_BVH_SBVH=0 HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi slots | grep -v "^=="
echo "#### standard tier 3840x2160, round-3 tree + slot order"; PB_ARGS="--width 3840 --height 2160" HR_BVH_SBVH=0 HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi slots | grep -v "^=="
} > $O/ab_hard_4k.txt 2>&1
