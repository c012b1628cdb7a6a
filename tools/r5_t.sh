#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5_t; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_tiling.py tests/test_gpu_comm.py tests/test_gpu_geo_history.py -q -x 2>&1 | tail -6
timeout 300 python tools/band_geo_ab.py 2>&1 | grep GEO | tee $OUT/band_geo_ab.txt
