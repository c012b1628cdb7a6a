#!/bin/bash
# round 6, GPU call AG: the fast DDGI sample redoes a shading point that lies ON a probe — the fuzz sequence that found it, the tolerance file, the cost
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ag; mkdir -p $OUT
cd $R
timeout 300 python tools/fuzz_one.py 6351 150 2>&1 | tail -2 | tee $OUT/fuzz_one_6351_150.txt
timeout 900 python -m pytest tests/test_gpu_tolerance.py tests/test_gpu_configs4.py tests/test_gpu_ddgi.py -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $OUT/pytest_subset.txt
for i in 1 2; do timeout 300 python tools/passbench.py --passes ddgi 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items()})
"; done | tee $OUT/passbench_ddgi.txt
