#!/bin/bash
# round 6, GPU call A: the six fuzz sequences that missed the 99.9 % population bound (docs/EXPERIMENTS.md R5.8) with the parity
# gather in the reflections trace (HR_REFL_FAST_SHADING=0), and what that gather costs (passbench, 1080p / 4K)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_a; mkdir -p $OUT
cd $R
for fs in 0 1; do
  for st in "31337 206" "555 66" "8088 21" "8088 61" "8088 84" "8088 159"; do
    echo "== HR_REFL_FAST_SHADING=$fs $st"; HR_REFL_FAST_SHADING=$fs timeout 600 python tools/fuzz_one.py $st 2>&1 | tail -2 | cut -c1-400
  done
done | tee $OUT/fuzz_one_six.txt
for res in "1920 1080" "3840 2160"; do
  set -- $res
  for fs in 1 0 1 0; do echo "#### $1x$2 HR_REFL_FAST_SHADING=$fs"; HR_REFL_FAST_SHADING=$fs timeout 600 python tools/passbench.py --width $1 --height $2 --passes reflections,ddgi 2>&1 | tail -4; done
done | tee $OUT/passbench_refl.txt
