#!/bin/bash
# round 6, GPU call I: whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_i; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
