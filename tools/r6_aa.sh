#!/bin/bash
# round 6, GPU call AA: range-branch-free quotients in the parity gather — parity, then passbench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_aa; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_reflections.py tests/test_gpu_ddgi.py tests/test_gpu_ref_shaders.py tests/test_gpu_configs.py tests/test_gpu_configs4.py tests/test_gpu_instances.py tests/test_gpu_textured.py tests/test_gpu_golden.py tests/test_gpu_post.py tests/test_gpu_tolerance.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/pytest_subset.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  for ex in 0 0 1; do echo "#### $1x$2 exact=$ex"; timeout 600 python tools/passbench.py --width $1 --height $2 --passes reflections,ddgi --exact $ex 2>&1 | tail -2 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items() if k in ('ray_trace', 'sample_probe_grid')})
"; done
done | tee $OUT/passbench_inrange_gather.txt
