#!/bin/bash
# round 6, GPU call Z: the driver's bench command with the instanced block; then the whole GPU suite on the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_z; mkdir -p $OUT
cd $R
(time python bench.py --gpus 1 --steps 20 --warmup 5) 2> $OUT/bench_driver_stderr.txt | tail -1 > $OUT/bench_driver_line.json; tail -4 $OUT/bench_driver_stderr.txt; wc -c $OUT/bench_driver_line.json
cp bench_detail.json $OUT/bench_detail.json
python -c "import json; d=json.load(open('$OUT/bench_driver_line.json')); print(d['value'], d['ms_per_step'], d['passes'].get('instanced'))"
timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | tee $OUT/pytest_gpu_full.txt | grep -E "passed|failed|error|skipped" | tail -6 | tee $OUT/pytest_gpu.txt
