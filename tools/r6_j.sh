#!/bin/bash
# round 6, GPU call J: whole GPU suite on the consolidated tree + the probe-table A/B of kf_ddgi_sample
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_j; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  for t in 1 0 1 0; do echo "#### $1x$2 HR_DDGI_PROBE_TABLE=$t"; HR_DDGI_PROBE_TABLE=$t timeout 600 python tools/passbench.py --width $1 --height $2 --passes ddgi 2>&1 | tail -1 | grep -o '"sample_probe_grid": {[^}]*}'; done
done | tee $OUT/ab_probe_table.txt
