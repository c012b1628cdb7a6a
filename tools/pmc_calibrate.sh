#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on kernels of known HBM bytes (tools/pmc_calib.hip); writes gpurun_out/<name>/pmc_calibration.json
NAME=${1:-calib}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ ! -x $R/tools/_build/pmc_calib ] || [ $R/tools/pmc_calib.hip -nt $R/tools/_build/pmc_calib ]; then
    mkdir -p $R/tools/_build && hipcc --offload-arch=gfx950 -O3 $R/tools/pmc_calib.hip -o $R/tools/_build/pmc_calib
fi
$R/tools/_build/pmc_calib > $OUT/calib_run.jsonl
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -- $R/tools/_build/pmc_calib > /dev/null 2> $OUT/cal_$c.err
done
python - "$OUT" <<'PY'
import sys, glob, csv, json, collections
out = sys.argv[1]
known = {json.loads(l)["kernel"]: json.loads(l) for l in open(out + "/calib_run.jsonl")}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/cal_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                acc[row["Kernel_Name"].split("(")[0]][c].append(float(row["Counter_Value"]))
res = {}
for k, cs in acc.items():
    kk = known.get(k, {})
    res[k] = dict(known_bytes=kk.get("known_bytes"), GBps_unprofiled=kk.get("GBps"))
    for c, v in cs.items():
        res[k][c + "_KB"] = sum(v) / len(v)
        if kk.get("known_bytes"):
            res[k][c + "_KB_x1024_over_known"] = round(sum(v) / len(v) * 1024 / kk["known_bytes"], 4)
json.dump(res, open(out + "/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE $OUT/*.err
