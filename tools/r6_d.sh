#!/bin/bash
# round 6, GPU call D: cost of hr_scene_update_instances and quality of the refitted tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_d; mkdir -p $OUT
cd $R
for cfg in "--detail 0.25 --movers 20" "--detail 1.0 --movers 200" "--detail 1.0 --movers 2000" "--detail 1.0 --movers 0"; do
  echo "#### $cfg"; timeout 900 python tools/instances_probe.py $cfg 2>&1 | tail -3
done | tee $OUT/instances_probe.txt
