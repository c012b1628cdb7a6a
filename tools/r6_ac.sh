#!/bin/bash
# round 6, GPU call AC: the fuzzers with new seeds on the final build (one-launch probe update, range-flag gather)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ac; mkdir -p $OUT
cd $R
timeout 1500 python tools/fuzz_instances.py 6300 120 > $OUT/fuzz_instances_6300x120.txt 2>&1; echo "fuzz_instances: $(tail -1 $OUT/fuzz_instances_6300x120.txt)"
timeout 2400 python tools/fuzz_tolerance.py 6301 1200 > $OUT/fuzz_tolerance_6301x1200.txt 2>&1; echo "fuzz_tolerance: $(tail -1 $OUT/fuzz_tolerance_6301x1200.txt)"
timeout 1200 python tools/fuzz_tolerance.py 6302 200 0 hard > $OUT/fuzz_hard_6302x200.txt 2>&1; echo "fuzz_hard: $(tail -1 $OUT/fuzz_hard_6302x200.txt)"
timeout 1800 python tools/fuzz_gpu.py 6303 200 > $OUT/fuzz_gpu_6303x200.txt 2>&1; echo "fuzz_gpu: $(tail -1 $OUT/fuzz_gpu_6303x200.txt)"
timeout 1200 python tools/fuzz_tiling.py 6304 120 > $OUT/fuzz_tiling_6304x120.txt 2>&1; echo "fuzz_tiling: $(tail -1 $OUT/fuzz_tiling_6304x120.txt)"
timeout 900 python tools/fuzz_bvh.py 6305 100 > $OUT/fuzz_bvh_6305x100.txt 2>&1; echo "fuzz_bvh: $(tail -1 $OUT/fuzz_bvh_6305x100.txt)"
timeout 900 python tools/fuzz_ddgi.py 6306 3000 > $OUT/fuzz_ddgi_6306x3000.txt 2>&1; echo "fuzz_ddgi: $(tail -1 $OUT/fuzz_ddgi_6306x3000.txt)"
