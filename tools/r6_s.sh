#!/bin/bash
# round 6, GPU call S: long fuzz campaigns on the final build (new seeds)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_s; mkdir -p $OUT
cd $R
timeout 1500 python tools/fuzz_instances.py 6100 120 > $OUT/fuzz_instances_6100x120.txt 2>&1; echo "fuzz_instances: $(tail -1 $OUT/fuzz_instances_6100x120.txt)"
timeout 2400 python tools/fuzz_tolerance.py 6001 1200 > $OUT/fuzz_tolerance_6001x1200.txt 2>&1; echo "fuzz_tolerance: $(tail -1 $OUT/fuzz_tolerance_6001x1200.txt)"
timeout 1200 python tools/fuzz_tolerance.py 6002 200 0 hard > $OUT/fuzz_hard_6002x200.txt 2>&1; echo "fuzz_hard: $(tail -1 $OUT/fuzz_hard_6002x200.txt)"
timeout 1800 python tools/fuzz_gpu.py 6003 200 > $OUT/fuzz_gpu_6003x200.txt 2>&1; echo "fuzz_gpu: $(tail -1 $OUT/fuzz_gpu_6003x200.txt)"
timeout 1200 python tools/fuzz_tiling.py 6004 120 > $OUT/fuzz_tiling_6004x120.txt 2>&1; echo "fuzz_tiling: $(tail -1 $OUT/fuzz_tiling_6004x120.txt)"
timeout 900 python tools/fuzz_bvh.py 6005 100 > $OUT/fuzz_bvh_6005x100.txt 2>&1; echo "fuzz_bvh: $(tail -1 $OUT/fuzz_bvh_6005x100.txt)"
