#!/bin/bash
# round 6, GPU call AD: the tolerance fuzz under the rule with the absolute floor — the seed that found the AO miss again, and a new one
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ad; mkdir -p $OUT
cd $R
timeout 2400 python tools/fuzz_tolerance.py 6301 1200 > $OUT/fuzz_tolerance_6301x1200.txt 2>&1; echo "6301: $(tail -1 $OUT/fuzz_tolerance_6301x1200.txt)"
timeout 2400 python tools/fuzz_tolerance.py 6311 1200 > $OUT/fuzz_tolerance_6311x1200.txt 2>&1; echo "6311: $(tail -1 $OUT/fuzz_tolerance_6311x1200.txt)"
