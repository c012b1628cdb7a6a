#!/bin/bash
# round 6, GPU call B: the whole GPU suite on the build that ships the parity gather in the reflections trace
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_b; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
