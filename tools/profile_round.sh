#!/bin/bash
# Produces the files of a profiles/<name>/ directory on the GPU box (under gpurun_out/<name>/):
#   bench.json               the JSON line of an un-profiled bench.py run (default flags)
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of the headline command (bench.py --no-passes --no-cpu-baseline)
#   frame_kernel_stats.csv   the same for every pass at 1080p (tools/passbench.py, tolerance mode): all kernels of the hybrid frame
#   pmc_summary.json         per-kernel average FETCH_SIZE / WRITE_SIZE (KB) per launch, one --pmc pass each, over the passbench run
#   sq_counters.txt          SQ instruction / busy counters per kernel (tools/pmc_sets.sh) over the passbench run
# --pmc passes never carry a trace domain (gpurun refuses the combination).
# usage: tools/profile_round.sh r2_a [exact]
#   PASS_ARGS="--width 3840 --height 2160" SKIP_BENCH=1 tools/profile_round.sh r3_4k    the hybrid frame's kernels at 4K only
NAME=${1:-round}
EXACT=${2:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=(python $R/bench.py --steps 100 --warmup 20 --exact $EXACT)
PASSES=(python $R/tools/passbench.py --exact $EXACT --frames 6 $PASS_ARGS)
if [ -z "$SKIP_BENCH" ]; then
python $R/bench.py --exact $EXACT 2>/dev/null | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- "${BENCH[@]}" --no-cpu-baseline --no-passes > /dev/null 2> $OUT/kt.err
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktf -- "${PASSES[@]}" > $OUT/passbench.json 2> $OUT/ktf.err
cp $(find $OUT/ktf -name '*kernel_stats.csv' | head -1) $OUT/frame_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- "${PASSES[@]}" > /dev/null 2> $OUT/pmc_$c.err
done
python - "$OUT" <<'EOF'
import sys, glob, csv, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(out + '/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == c:
                acc[row['Kernel_Name']][c].append(float(row['Counter_Value']))
summary = {}
for k, cs in acc.items():
    summary[k] = {}
    for c, v in cs.items():
        summary[k][c + '_KB_avg_per_launch'] = round(sum(v) / len(v), 1)
        summary[k]['launches_' + c] = len(v)
json.dump(summary, open(out + '/pmc_summary.json', 'w'), indent=1)
EOF
bash $R/tools/pmc_sets.sh $OUT/sq -- "${PASSES[@]}" > $OUT/sq_counters.txt 2>&1
rm -rf $OUT/kt $OUT/ktf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq $OUT/*.err
ls -la $OUT
