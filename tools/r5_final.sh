#!/bin/bash
# round 5, final GPU call: the whole GPU suite on the final build + the bench records that read profiles/r5_c
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_final; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json; cp bench_detail.json $OUT/bench_detail.json; wc -c $OUT/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver.json; wc -c $OUT/bench_driver.json
