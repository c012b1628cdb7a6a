#!/bin/bash
# round 6, GPU call C: instanced scenes — parity tests, then what an update costs (tools/instances_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_c; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_instances.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_instances.txt
