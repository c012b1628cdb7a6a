#!/bin/bash
# round 6, GPU call H: whole GPU suite (markers, merged gather site, instanced scenes, bench line) + the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_h; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
(time python bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench_line.txt 2> $OUT/bench_stderr.txt; tail -c 600 $OUT/bench_stderr.txt
tail -1 $OUT/bench_line.txt | wc -c
cp bench_detail.json $OUT/ 2>/dev/null
