#!/bin/bash
# Round 6, GPU call 4: R7 with the next batch's list words prefetched (A/B on the S-map and on the trained map); tests of the loss hoist + the sharded fix.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mapper_ops_gpu.py tests/test_sharded_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2
GSICP_BWD_PREFETCH=1 timeout 900 python -m pytest tests/test_raster_gpu.py -m gpu -x -q -k "backward or trained or full_size" > $OUT/pytest_prefetch.log 2>&1
grep -E "passed|failed" $OUT/pytest_prefetch.log | tail -2
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
T="python $ROOT/bench.py --only trained --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_t.json"
for v in 0 1; do
  GSICP_BWD_PREFETCH=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_smap_pf$v -o bench -- $M > $OUT/smap_pf$v.json 2> $OUT/kt_smap_pf$v.err
  GSICP_BWD_PREFETCH=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained_pf$v -o bench -- $T > $OUT/trained_pf$v.json 2> $OUT/kt_trained_pf$v.err
  GSICP_BWD_PREFETCH=$v $M > $OUT/smap_plain_pf$v.json 2>> $OUT/kt_smap_pf$v.err
done
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for d in smap_pf0 smap_pf1 trained_pf0 trained_pf1; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/kt_$d/bench_kernel_stats.csv")))[:6]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
for v in (0,1):
    d=json.load(open('$OUT/smap_plain_pf%d.json'%v)); print('smap mapper_only pf',v, d['ms_per_step'])
    d=json.load(open('$OUT/trained_pf%d.json'%v)); print('trained (under rocprof) pf',v, d['ms_per_step'])
"
