#!/bin/bash
# round 6, GPU call R: narrow refit levels in one launch — parity, then the update cost
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_r; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_instances.py tests/test_gpu_cpp_example.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_instances.txt
for cfg in "--detail 0.25 --movers 20" "--detail 1.0 --movers 200" "--detail 1.0 --movers 2000" "--detail 1.0 --movers 2000 --frames 120"; do
  echo "#### $cfg"; timeout 900 python tools/instances_probe.py $cfg 2>&1 | tail -1
done | tee $OUT/instances_probe.txt
