#!/bin/bash
# round 6, GPU call AF: more tolerance sequences under the final rule
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_af; mkdir -p $OUT
cd $R
timeout 2400 python tools/fuzz_tolerance.py 6331 1200 > $OUT/fuzz_tolerance_6331x1200.txt 2>&1; echo "6331: $(tail -1 $OUT/fuzz_tolerance_6331x1200.txt)"
timeout 2400 python tools/fuzz_tolerance.py 6341 1200 > $OUT/fuzz_tolerance_6341x1200.txt 2>&1; echo "6341: $(tail -1 $OUT/fuzz_tolerance_6341x1200.txt)"
timeout 1500 python tools/fuzz_tolerance.py 6342 300 0 hard > $OUT/fuzz_hard_6342x300.txt 2>&1; echo "hard 6342: $(tail -1 $OUT/fuzz_hard_6342x300.txt)"
