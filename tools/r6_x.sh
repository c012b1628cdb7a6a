#!/bin/bash
# round 6, GPU call X: probe update + borders as one launch — parity (DDGI atlases through every suite that compares them), then passbench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_x; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ddgi.py tests/test_gpu_ref_shaders.py tests/test_gpu_configs.py tests/test_gpu_configs4.py tests/test_gpu_reflections.py tests/test_gpu_shadows.py tests/test_gpu_tiling.py tests/test_gpu_comm.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_tolerance.py -x -q > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 | tee $OUT/pytest_subset.txt
for ex in 0 0 1; do echo "#### 1920x1080 exact=$ex"; timeout 600 python tools/passbench.py --width 1920 --height 1080 --passes ddgi --exact $ex 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(' ', d['pass_'], d['ms_per_frame'], {k: v['ms'] for k, v in d['stages'].items()})
"; done | tee $OUT/passbench_fused_probe_update.txt
