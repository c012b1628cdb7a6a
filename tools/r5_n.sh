#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_n; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_configs4.py -q --tb=short -k "4k_ddgi or composite" 2>&1 | grep -E "^E|Error|passed|failed" | cut -c1-400 | head -30
export PB_ARGS="--width 1920 --height 1080"; bash tools/ab.sh shadows,ao r5base base r5base base 2>&1 | sed -E "s/'atrous_01.*//; s/'blur.*//"
( time timeout 1500 python tools/fuzz_tolerance.py 501 200 ) > $OUT/fuzz_tolerance_501.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_501.txt | head -1
( time timeout 900 python tools/fuzz_tolerance.py 31 20 20 ) > $OUT/fuzz_tolerance_long.txt 2>&1; tail -n 4 $OUT/fuzz_tolerance_long.txt | head -1
grep -h "OUT OF\|ERROR" $OUT/fuzz_tolerance_*.txt | cut -c1-300
