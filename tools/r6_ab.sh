#!/bin/bash
# round 6, GPU call AB: AO trace with the blue-noise texel hoisted and the Sobol bytes prefetched — parity, then passbench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ab; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ao.py tests/test_gpu_ref_shaders.py tests/test_gpu_configs.py tests/test_gpu_configs4.py tests/test_gpu_golden.py tests/test_gpu_edge.py tests/test_gpu_instances.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest_subset.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  for i in 1 2 3; do timeout 600 python tools/passbench.py --width $1 --height $2 --passes ao 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', '$1x$2', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items()})
"; done
done | tee $OUT/passbench_ao_prefetch.txt
