#!/bin/bash
# Round 6, GPU call 8: tile sort's bitonic index arithmetic with shifts / masks instead of integer divisions: raster parity tests + kernel traces (S-map, trained map).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06h
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -1
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
T="python $ROOT/bench.py --only trained --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_t.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_smap -o bench -- $M > $OUT/smap.json 2> $OUT/kt_smap.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/trained.json 2> $OUT/kt_trained.err
$M > $OUT/smap_plain.json 2>> $OUT/kt_smap.err
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for d in smap trained; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/kt_$d/bench_kernel_stats.csv")))[:7]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
d=json.load(open('$OUT/smap_plain.json')); print('smap mapper_only', d['ms_per_step'])
d=json.load(open('$OUT/trained.json')); print('trained (under rocprof)', d['ms_per_step'])"
