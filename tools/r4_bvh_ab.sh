#!/bin/bash
# round 4, GPU: the new BVH build (spatial splits + reinsertion) and the far-to-near any-hit order against what round 3 shipped
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_a; mkdir -p $O
cd $R
export FRAMES=30
{
echo "#### new tree, any-hit far-first (product)";            bash tools/ab.sh shadows,ao,reflections,ddgi base
echo "#### new tree, any-hit slot order";                     bash tools/ab.sh shadows,ao,reflections,ddgi slots
echo "#### new tree, AO far-first too";                       bash tools/ab.sh ao aofar
echo "#### round-3 tree (no spatial splits, no reinsertion), far-first"; HR_BVH_SBVH=0 HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi base
echo "#### round-3 tree, slot order (= round 3)";             HR_BVH_SBVH=0 HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi slots
echo "#### spatial splits only";                              HR_BVH_REINSERT=0 bash tools/ab.sh shadows,ao,reflections,ddgi base
echo "#### reinsertion only";                                 HR_BVH_SBVH=0 bash tools/ab.sh shadows,ao,reflections,ddgi base
} > $O/ab_1080p.txt 2>&1
