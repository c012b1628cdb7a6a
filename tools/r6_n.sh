#!/bin/bash
# round 6, GPU call N: the rest of round 5's 1214-sequence campaign (same seeds and counts) under the shipped rule, + bit-exact and tiling fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_n; mkdir -p $OUT
cd $R
for sc in "7777 200" "31337 400" "777 100"; do set -- $sc; timeout 1800 python tools/fuzz_tolerance.py $1 $2 > $OUT/fuzz_tolerance_$1x$2.txt 2>&1; echo "$1 x $2: $(tail -1 $OUT/fuzz_tolerance_$1x$2.txt)"; done
for sc in "4242 60" "99 100"; do set -- $sc; timeout 1800 python tools/fuzz_tolerance.py $1 $2 0 hard > $OUT/fuzz_hard_$1x$2.txt 2>&1; echo "hard $1 x $2: $(tail -1 $OUT/fuzz_hard_$1x$2.txt)"; done
timeout 1200 python tools/fuzz_gpu.py 6006 80 > $OUT/fuzz_gpu_6006x80.txt 2>&1; echo "fuzz_gpu: $(tail -1 $OUT/fuzz_gpu_6006x80.txt)"
timeout 900 python tools/fuzz_tiling.py 6006 50 > $OUT/fuzz_tiling_6006x50.txt 2>&1; echo "fuzz_tiling: $(tail -1 $OUT/fuzz_tiling_6006x50.txt)"
