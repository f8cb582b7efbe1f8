#!/bin/bash
# Round 6, GPU call 9: the tile sort's wave-local steps in registers: parity tests + kernel traces (S-map, trained map), A/B against the LDS network.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06i
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_raster_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^E " $OUT/pytest.log | head -5
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
T="python $ROOT/bench.py --only trained --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_t.json"
for v in 0 1; do
  GSICP_TILE_SORT_LDS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_smap_lds$v -o bench -- $M > $OUT/smap_lds$v.json 2> $OUT/kt_smap_lds$v.err
  GSICP_TILE_SORT_LDS=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained_lds$v -o bench -- $T > $OUT/trained_lds$v.json 2> $OUT/kt_trained_lds$v.err
  GSICP_TILE_SORT_LDS=$v $M > $OUT/smap_plain_lds$v.json 2>> $OUT/kt_smap_lds$v.err
done
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for d in smap_lds0 smap_lds1 trained_lds0 trained_lds1; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/kt_$d/bench_kernel_stats.csv")))[:8]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    if m and m.group(1) in ("tile_sort_kernel", "blend_forward_strip_kernel"): print("%-34s calls %5s avg_us %9.2f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
for v in (0,1):
    d=json.load(open('$OUT/smap_plain_lds%d.json'%v)); print('smap mapper_only lds',v, d['ms_per_step'])
    d=json.load(open('$OUT/trained_lds%d.json'%v)); print('trained (under rocprof) lds',v, d['ms_per_step'])"
