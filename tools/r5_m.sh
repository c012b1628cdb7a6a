#!/bin/bash
# round 5, GPU call M: the whole GPU suite + fuzz campaigns under the SHIPPED tolerance rule (no env overrides) on the committed build
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_m; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_tolerance_501.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_501.txt
( time timeout 1500 python tools/fuzz_tolerance.py 2025 150 ) > $OUT/fuzz_tolerance_2025.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_2025.txt
( time timeout 900 python tools/fuzz_tolerance.py 31 20 20 ) > $OUT/fuzz_tolerance_long.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_long.txt
grep -h "OUT OF\|ERROR" $OUT/fuzz_tolerance_*.txt | cut -c1-300
