#!/bin/bash
# Round 5, GPU call 3: tests of the mapper-side changes (run summation, select_view, in-kernel step bump, row freeze, policy) + kernel traces.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_graph_gpu.py tests/test_mapper_ops_gpu.py tests/test_sharded_gpu.py tests/test_store_gpu.py tests/test_slam_loop_gpu.py -q > $OUT/pytest_mapper.log 2>&1
tail -25 $OUT/pytest_mapper.log | cut -c1-400
timeout 900 python -m pytest tests/test_reference_slam_gpu.py -q -k "noise or freezes or fused" > $OUT/pytest_refslam.log 2>&1
tail -25 $OUT/pytest_refslam.log | cut -c1-600
grep -h "noisy \(replica\|tum\)" $OUT/pytest_refslam.log | cut -c1-400
cd /tmp
T="python $ROOT/bench.py --only trained --steps 50 --repeats 2"
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/trained_leg.json 2> $OUT/kt_trained.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/mapper_only.json 2> $OUT/kt_mapper.err
$M > $OUT/mapper_only_plain.json 2>> $OUT/kt_mapper.err
cd $ROOT
find $OUT -name '*.csv' -size +20M -delete
for d in kt_trained kt_mapper; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/$d/bench_kernel_stats.csv")))[:18]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
for f in ('mapper_only_plain','trained_leg'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['ms_per_step'])"
