#!/bin/bash
# Produces the files of a profiles/<name>/ directory on the GPU box (under gpurun_out/<name>/):
#   bench.json                   the JSON line of an un-profiled bench.py run (default flags)
#   kernel_stats.csv             rocprofv3 --kernel-trace --stats of the headline command (bench.py --no-passes --no-cpu-baseline)
#   frame_kernel_stats[_4k].csv  the same for every pass of the hybrid frame (tools/passbench.py, tolerance mode) at 1920x1080 / 3840x2160
#   pmc_summary[_4k].json        per-kernel average FETCH_SIZE / WRITE_SIZE (KB) per launch, one --pmc pass each, over the passbench run
#   sq_counters[_4k].json/.txt   raw SQ / GRBM counters per kernel (tools/pmc_sets.sh) over the passbench run
# --pmc passes never carry a trace domain (gpurun refuses the combination).  bench.py reads these files by EXACT kernel name.
# usage: tools/profile_round.sh r3_b [exact]          SKIP_BENCH=1 / SKIP_4K=1 / SKIP_SQ=1 shorten the run
NAME=${1:-round}
EXACT=${2:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH=(python $R/bench.py --steps 100 --warmup 20 --exact $EXACT)
if [ -z "$SKIP_BENCH" ]; then
    python $R/bench.py --exact $EXACT 2>/dev/null | tail -1 > $OUT/bench.json          # the compact stdout line (what the driver parses)
    cp $R/bench_detail.json $OUT/bench_detail.json 2>/dev/null                          # the full record (per-kernel blocks, notes)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- "${BENCH[@]}" --no-cpu-baseline --no-passes > /dev/null 2> $OUT/kt.err
    cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
fi
for cfg in "1920 1080 _" "3840 2160 _4k"; do
    set -- $cfg
    SUF=${3#_}; [ -n "$SUF" ] && SUF=_$SUF
    [ -n "$SUF" ] && [ -n "$SKIP_4K" ] && continue
    PASSES=(python $R/tools/passbench.py --exact $EXACT --frames 6 --width $1 --height $2 $PASS_ARGS)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktf -- "${PASSES[@]}" > $OUT/passbench$SUF.json 2> $OUT/ktf.err
    cp $(find $OUT/ktf -name '*kernel_stats.csv' | head -1) $OUT/frame_kernel_stats$SUF.csv
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- "${PASSES[@]}" > /dev/null 2> $OUT/pmc_$c.err
    done
    python - "$OUT" "$SUF" <<'PY'
import sys, glob, csv, json, collections
out, suf = sys.argv[1], sys.argv[2]
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
json.dump(summary, open(out + '/pmc_summary%s.json' % suf, 'w'), indent=1)
PY
    if [ -z "$SKIP_SQ" ]; then
        bash $R/tools/pmc_sets.sh $OUT/sq -- "${PASSES[@]}" > $OUT/sq_counters$SUF.txt 2>&1
        cp $OUT/sq/sq_counters.json $OUT/sq_counters$SUF.json
    fi
    rm -rf $OUT/ktf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/sq
done
rm -rf $OUT/kt $OUT/*.err
ls -la $OUT
