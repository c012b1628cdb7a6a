#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_p; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_configs4.py tests/test_gpu_configs.py -q --tb=short 2>&1 | grep -E "^E|Error|passed|failed" | cut -c1-600 | head -30
export PB_ARGS="--width 3840 --height 2160"; bash tools/ab.sh ddgi r5base base 2>&1
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_tolerance_501.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_501.txt | head -1
grep -h "OUT OF\|ERROR" $OUT/fuzz_tolerance_*.txt | cut -c1-300
