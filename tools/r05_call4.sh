#!/bin/bash
# Round 5, GPU call 4: backward without FMA contraction (legacy == new bit for bit), unrolled run-sum folds, Adam bump A/B, new tracker tests, bench line.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05e
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_graph_gpu.py -q > $OUT/pytest_mapper.log 2>&1
tail -12 $OUT/pytest_mapper.log | cut -c1-300
grep -n "AssertionError" $OUT/pytest_mapper.log | cut -c1-400 | head
timeout 600 python -m pytest tests/test_gicp_gpu.py -q -k "second_index or tum_configuration" > $OUT/pytest_gicp_new.log 2>&1
tail -8 $OUT/pytest_gicp_new.log | cut -c1-300
cd /tmp
T="python $ROOT/bench.py --only trained --steps 50 --repeats 2"
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/trained_leg.json 2> $OUT/kt_trained.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/mapper_only.json 2> $OUT/kt_mapper.err
GSICP_ADAM_INKERNEL_BUMP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_inkernel_bump -o bench -- $M > $OUT/mapper_only_inkernel_bump.json 2> $OUT/kt_mapper_b.err
$M > $OUT/mapper_only_plain.json 2>> $OUT/kt_mapper.err
GSICP_EMIT_WALK=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_emit_walk -o bench -- $M > $OUT/mapper_only_emit_walk.json 2> $OUT/kt_mapper_w.err
cd $ROOT
find $OUT -name '*.csv' -size +20M -delete
for d in kt_trained kt_mapper kt_mapper_inkernel_bump kt_mapper_emit_walk; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/$d/bench_kernel_stats.csv")))[:17]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
for f in ('mapper_only_plain','trained_leg'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, d['ms_per_step'], d.get('value'))
"
