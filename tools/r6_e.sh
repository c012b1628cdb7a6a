#!/bin/bash
# round 6, GPU call E: instanced scenes with leaf cells + dirty-instance updates: parity, then cost / quality probe
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_e; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_instances.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_instances.txt
python tools/_bvh_cmp.py 2>&1 | head -6 | tee $OUT/bvh_cmp.txt
for cfg in "--detail 0.25 --movers 20" "--detail 1.0 --movers 200" "--detail 1.0 --movers 2000" "--detail 1.0 --movers 0"; do
  echo "#### $cfg"; timeout 900 python tools/instances_probe.py $cfg 2>&1 | tail -3
done | tee $OUT/instances_probe.txt
