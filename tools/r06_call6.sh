#!/bin/bash
# Round 6, GPU call 6: the pre-zeroed captured forward (counters cleared by the keyframe-selection launch): tests, A/B trace, headline.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06f
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_sharded_gpu.py tests/test_sharded_2rank_gpu.py tests/test_bench_gpu.py tests/test_slam_loop_gpu.py tests/test_store_gpu.py tests/test_mapper_ops_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 900 python -m pytest tests/test_reference_slam_gpu.py -m gpu -x -q -k "fused_rows or fused_iteration or fused_gaussian" > $OUT/pytest_ref.log 2>&1
grep -E "passed|failed" $OUT/pytest_ref.log | tail -2
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
for v in 1 0; do
  GSICP_PREZERO=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_prezero$v -o bench -- $M > $OUT/mapper_only_prezero$v.json 2> $OUT/kt_prezero$v.err
  GSICP_PREZERO=$v $M > $OUT/mapper_only_plain_prezero$v.json 2>> $OUT/kt_prezero$v.err
  GSICP_PREZERO=$v python $ROOT/bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --legs-file /tmp/legs_h.json > $OUT/bench_prezero$v.json 2>> $OUT/kt_prezero$v.err
done
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for v in 1 0; do echo == prezero $v; python - <<PY
import csv, re, json
tot=0
for r in list(csv.DictReader(open("$OUT/kt_prezero$v/bench_kernel_stats.csv")))[:18]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    n = m.group(1) if m else r["Name"][:30]
    if int(r["Calls"]) > 1000: tot += float(r["AverageNs"]) / 1e3
    if n in ("zero_fill_kernel", "select_view_kernel", "select_view_zero_kernel", "preprocess_kernel"): print("%-34s calls %5s avg_us %9.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
print("sum of kernels with > 1000 calls", round(tot, 1))
d=json.load(open("$OUT/mapper_only_plain_prezero$v.json")); print('mapper_only plain', d['ms_per_step'])
d=json.load(open("$OUT/bench_prezero$v.json")); print('headline', d['ms_per_step'], d['block_ms_per_step_p10_p50_p90'])
PY
done
