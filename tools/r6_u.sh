#!/bin/bash
# round 6, GPU call U: the new seeds again under the per-iteration stage-wise check
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_u; mkdir -p $OUT
cd $R
timeout 2400 python tools/fuzz_tolerance.py 6001 1200 > $OUT/fuzz_tolerance_6001x1200.txt 2>&1; echo "fuzz_tolerance: $(tail -1 $OUT/fuzz_tolerance_6001x1200.txt)"
timeout 1200 python tools/fuzz_tolerance.py 6002 200 0 hard > $OUT/fuzz_hard_6002x200.txt 2>&1; echo "fuzz_hard: $(tail -1 $OUT/fuzz_hard_6002x200.txt)"
timeout 1200 python tools/fuzz_tolerance.py 8088 160 > $OUT/fuzz_tolerance_8088x160.txt 2>&1; echo "8088: $(tail -1 $OUT/fuzz_tolerance_8088x160.txt)"
