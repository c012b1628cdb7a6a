#!/bin/bash
# round 6, GPU call M: tolerance fuzz campaign under the shipped rule (tools/fuzz_tolerance.py): the seeds whose round-5 runs held the six missing sequences
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_m; mkdir -p $OUT
cd $R
timeout 2400 python tools/fuzz_tolerance.py 8088 160 > $OUT/fuzz_tolerance_8088x160.txt 2>&1; tail -1 $OUT/fuzz_tolerance_8088x160.txt
timeout 2400 python tools/fuzz_tolerance.py 555 200 > $OUT/fuzz_tolerance_555x200.txt 2>&1; tail -1 $OUT/fuzz_tolerance_555x200.txt
timeout 2400 python tools/fuzz_tolerance.py 31337 210 > $OUT/fuzz_tolerance_31337x210.txt 2>&1; tail -1 $OUT/fuzz_tolerance_31337x210.txt
