#!/bin/bash
# Round 5, GPU call 5: A/B of the tile-scan merge (last workgroup of the column scan) against the separate launch.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05f
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/mapper_only.json 2> $OUT/kt_mapper.err
GSICP_TILE_SCAN_LAUNCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_scan_launch -o bench -- $M > $OUT/mapper_only_scan_launch.json 2> $OUT/kt_mapper_s.err
$M > $OUT/mapper_only_plain.json 2>> $OUT/kt_mapper.err
GSICP_TILE_SCAN_LAUNCH=1 $M > $OUT/mapper_only_plain_scan_launch.json 2>> $OUT/kt_mapper.err
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for d in kt_mapper kt_mapper_scan_launch; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/$d/bench_kernel_stats.csv")))[:18]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
for f in ('mapper_only_plain','mapper_only_plain_scan_launch'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['ms_per_step'], d.get('value'))
"
