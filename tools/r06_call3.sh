#!/bin/bash
# Round 6, GPU call 3: loss-kernel forms (A/B traces, bit-identity tests) and the CU-split experiment (tracker on dedicated CUs).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mapper_ops_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log | cut -c1-300
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
for v in 0 1 2 3; do
  GSICP_LOSS_TILE3=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_loss_$v -o bench -- $M > $OUT/mapper_only_loss_$v.json 2> $OUT/kt_loss_$v.err
  GSICP_LOSS_TILE3=$v $M > $OUT/mapper_only_plain_loss_$v.json 2>> $OUT/kt_loss_$v.err
done
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for v in 0 1 2 3; do echo == loss form $v; python - <<PY
import csv, re, json
for r in list(csv.DictReader(open("$OUT/kt_loss_$v/bench_kernel_stats.csv")))[:8]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    if m and 'loss' in m.group(1): print("%-34s calls %5s avg_us %9.2f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
d=json.load(open("$OUT/mapper_only_plain_loss_$v.json")); print('mapper_only plain', d['ms_per_step'])
PY
done
# CU split: headline step, both halves co-tenant
for rep in 1 2; do
for n in 0 32 48 64 96; do
  python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --cu-split $n --legs-file $OUT/legs_cu_${n}_$rep.json > $OUT/bench_cu_${n}_$rep.json 2>> $OUT/cu.err
  python -c "
import json; d=json.load(open('$OUT/bench_cu_${n}_$rep.json')); print('cu-split $n rep $rep: ms_per_step', d['ms_per_step'], 'p10/50/90', d['block_ms_per_step_p10_p50_p90'], 'align us', d.get('tracker_align_kernel_us'))"
done
done
for n in 0 48; do
  python bench.py --only tracker --steps 50 --warmup 5 --no-legs --no-cpu-baseline --cu-split $n --legs-file $OUT/legs_trk_$n.json > $OUT/bench_trk_$n.json 2>> $OUT/cu.err
  python bench.py --only mapper --steps 50 --warmup 5 --no-legs --no-cpu-baseline --cu-split $n --legs-file $OUT/legs_map_$n.json > $OUT/bench_map_$n.json 2>> $OUT/cu.err
  python -c "
import json; a=json.load(open('$OUT/bench_trk_$n.json')); b=json.load(open('$OUT/bench_map_$n.json')); print('cu-split $n alone: tracker', a['ms_per_step'], 'mapper', b['ms_per_step'])"
done
tail -3 $OUT/cu.err
