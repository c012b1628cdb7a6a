#!/bin/bash
# round 6, GPU call L: SAH top level + automatic re-build: parity, then the probe with / without re-builds
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_l; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_instances.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_instances.txt
for rb in 1 0; do
for cfg in "--detail 1.0 --movers 200" "--detail 1.0 --movers 2000" "--detail 1.0 --movers 2000 --frames 120"; do
  echo "#### HR_TOP_LEVEL_REBUILD=$rb $cfg"; HR_TOP_LEVEL_REBUILD=$rb timeout 900 python tools/instances_probe.py $cfg 2>&1 | tail -1
done; done | tee $OUT/instances_probe.txt
