#!/bin/bash
# round 5, GPU call W: long runs + two more seeds under the shipped tolerance rule; the switch matrix of the GPU suite's core tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_w; mkdir -p $OUT
cd $R
( time timeout 1500 python tools/long_run_tolerance.py 120 ) > $OUT/long_run_tolerance.txt 2>&1; tail -n 5 $OUT/long_run_tolerance.txt | head -2
( time timeout 1500 python tools/fuzz_tolerance.py 9001 200 ) > $OUT/fuzz_tolerance_9001.txt 2>&1; tail -n 5 $OUT/fuzz_tolerance_9001.txt | head -1
grep -h "OUT OF\|ERROR" $OUT/*.txt | cut -c1-300
for sw in "HR_TILE_ORDER=0" "HR_GEO_HISTORY=0" "HR_SHADOW_CACHE=0" "HR_TILE_ORDER_FUSED=0"; do
  echo "== $sw"; env $sw timeout 900 python -m pytest tests/test_gpu_shadows.py tests/test_gpu_ao.py tests/test_gpu_tolerance.py tests/test_gpu_tiling.py -q -x 2>&1 | tail -1
done | tee $OUT/switch_matrix.txt
