#!/bin/bash
# round 6, GPU call T: the per-iteration stage-wise check on the four sequences that failed the chain check + the tolerance tests; instance fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_t; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_tolerance.py -x -q 2>&1 | tail -4 | tee $OUT/pytest_tolerance.txt
python - <<'PY' 2>&1 | tee $OUT/four_sequences.txt
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from hybrid_rendering_amd import api as hr
from oracle import pyoracle as oracle
import test_gpu_tolerance as tol, helpers
ctx = hr.Context(0)
for seed, trial, hard in ((6001, 141, False), (6002, 9, True), (6002, 97, True), (6002, 98, True)):
    c = helpers.fuzz_config(seed, trial, hard)
    try:
        tol.test_reflections_and_ddgi_sample_tolerance(oracle, hr, ctx, c["name"], c["W"], c["H"], min(c["scale"], 1), c["dolly"], c["reflections"])
        print(seed, trial, "ok")
    except AssertionError as e:
        print(seed, trial, "FAILED:", str(e)[:300])
PY
timeout 1500 python tools/fuzz_instances.py 6100 120 > $OUT/fuzz_instances_6100x120.txt 2>&1; echo "fuzz_instances: $(tail -1 $OUT/fuzz_instances_6100x120.txt)"
