#!/bin/bash
# round 5, GPU call V: bit-exact fuzz campaigns on the final build + the final bench record (counters = profiles/r5_b)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_v; mkdir -p $OUT
cd $R
( time timeout 1200 python tools/fuzz_gpu.py 5005 120 ) > $OUT/fuzz_gpu.txt 2>&1; tail -n 5 $OUT/fuzz_gpu.txt | head -2
( time timeout 900 python tools/fuzz_tiling.py 5006 60 ) > $OUT/fuzz_tiling.txt 2>&1; tail -n 5 $OUT/fuzz_tiling.txt | head -2
( time timeout 600 python tools/fuzz_bvh.py 5007 80 --gpu ) > $OUT/fuzz_bvh.txt 2>&1; tail -n 5 $OUT/fuzz_bvh.txt | head -2
python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json; cp bench_detail.json $OUT/bench_detail.json; wc -c $OUT/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver.json; wc -c $OUT/bench_driver.json
