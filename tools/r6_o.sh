#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_o; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_instances.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_instances.txt
