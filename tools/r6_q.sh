#!/bin/bash
# round 6, GPU call Q: bench.py as the driver runs it, reading the counters of profiles/r6_f; smoke(); the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_q; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
(time python bench.py --gpus 1 --steps 20 --warmup 5) 2> $OUT/bench_driver_stderr.txt | tail -1 > $OUT/bench_driver_line.json; tail -4 $OUT/bench_driver_stderr.txt
cp bench_detail.json $OUT/bench_detail_driver.json
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json; cp bench_detail.json $OUT/bench_detail.json
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
