#!/bin/bash
# round 6, GPU call AH: the parity second run for knife-edge history taps in kf_refl_temporal — the open sequence, the tolerance / tiling / fused tests, the cost
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_ah; mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_gpu_tolerance.py tests/test_gpu_fused.py tests/test_gpu_tiling.py tests/test_gpu_reflections.py -q -rxX 2>&1 | grep -E "passed|failed|error|XPASS|XFAIL" | tail -4 | tee $OUT/pytest_subset.txt
for res in "1920 1080" "3840 2160"; do set -- $res; timeout 100 python tools/passbench.py --width $1 --height $2 --passes reflections 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', '$1x$2', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items()})
"; done | tee $OUT/passbench_refl.txt
