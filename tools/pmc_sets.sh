#!/bin/bash
# Collects SQ / TA / L2 counters for bench.py (or any command given as arguments) in separate rocprofv3 --pmc
# passes (no trace domains next to --pmc) and prints the per-kernel averages.
#   tools/pmc_sets.sh [outdir] -- cmd...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sets
if [ "$1" != "--" ] && [ -n "$1" ]; then OUT=$1; shift; fi
[ "$1" == "--" ] && shift
CMD=("$@")
[ ${#CMD[@]} -eq 0 ] && CMD=(python $R/bench.py --steps 3 --warmup 1)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- "${CMD[@]}" > $OUT/set$i.out 2> $OUT/set$i.err || echo "set$i failed: $(tail -2 $OUT/set$i.err)"
done
python - "$OUT" <<'EOF'
import sys, glob, csv, collections, json, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/set*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in sorted(acc.items()):
    print(k[:110])
    for c, v in sorted(cs.items()):
        print('   %-28s avg %14.1f  (n=%d)' % (c, sum(v) / len(v), len(v)))
# raw per-launch averages by full kernel name (bench.py: valu_issue_frac, lane_utilisation)
json.dump({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if re.sub(r'^void\s+', '', k).startswith(('k_', 'kf_', '(anonymous'))},
          open(out + '/sq_counters.json', 'w'), indent=1)
EOF
