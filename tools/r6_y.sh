#!/bin/bash
# round 6, GPU call Y: the new DDGI grid-shape test, then the final profile round (profiles/r6_h)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_h; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_ddgi.py -x -q > $OUT/pytest_ddgi.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_ddgi.txt | tail -3
timeout 2400 bash tools/profile_round.sh r6_h > $OUT/profile_round.log 2>&1
tail -3 $OUT/profile_round.log
