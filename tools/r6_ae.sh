#!/bin/bash
# round 6, GPU call AE: the tolerance fuzz under the final rule (floor + the shadows chain's allowance, stage-wise shadows a-trous check in the runner)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ae; mkdir -p $OUT
cd $R
timeout 2400 python tools/fuzz_tolerance.py 6311 1200 > $OUT/fuzz_tolerance_6311x1200.txt 2>&1; echo "6311: $(tail -1 $OUT/fuzz_tolerance_6311x1200.txt)"
timeout 2400 python tools/fuzz_tolerance.py 6321 1200 > $OUT/fuzz_tolerance_6321x1200.txt 2>&1; echo "6321: $(tail -1 $OUT/fuzz_tolerance_6321x1200.txt)"
