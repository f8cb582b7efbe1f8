#!/bin/bash
# Round 6, GPU call 2: tests of sparse gradient rows + the step bump in the loss + the compact bench line; A/B kernel traces; the driver's command, timed.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_graph_gpu.py tests/test_mapper_ops_gpu.py tests/test_bench_gpu.py tests/test_sharded_gpu.py tests/test_sharded_2rank_gpu.py tests/test_store_gpu.py tests/test_slam_loop_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log | cut -c1-400
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err ) 2> $OUT/bench_driver_cmd.time
cat $OUT/bench_driver_cmd.time | tail -4
cp bench_legs.json $OUT/bench_legs_driver_cmd.json 2>/dev/null
wc -c $OUT/bench_driver_cmd.json
python - <<PY
import json
d=json.load(open("$OUT/bench_driver_cmd.json"))
print({k: d.get(k) for k in ("value","ms_per_step","render_bwd_ms_per_iter","loss_adam_ms_per_iter","mapper_only_ms_per_iter","tracker_only_ms_per_frame","mapper_iteration_ms_trained_map","system_fps","ate_cm","section_wall_s")})
print("roofline", d["roofline"]); print("cpu", d["cpu_baseline"])
f=json.load(open("$OUT/bench_legs_driver_cmd.json")); print("sections", f.get("section_wall_s"))
PY
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/legs_m.json"
for v in new old; do
  if [ $v = old ]; then export GSICP_SPARSE_GRADS=0 GSICP_STEP_BUMP_IN_LOSS=0; else unset GSICP_SPARSE_GRADS GSICP_STEP_BUMP_IN_LOSS; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_$v -o bench -- $M > $OUT/mapper_only_$v.json 2> $OUT/kt_mapper_$v.err
  $M > $OUT/mapper_only_plain_$v.json 2>> $OUT/kt_mapper_$v.err
done
unset GSICP_SPARSE_GRADS GSICP_STEP_BUMP_IN_LOSS
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for v in new old; do echo == $v; python - <<PY
import csv, re, json
tot=0
for r in list(csv.DictReader(open("$OUT/kt_mapper_$v/bench_kernel_stats.csv")))[:22]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
d=json.load(open("$OUT/mapper_only_plain_$v.json")); print('mapper_only plain', d['ms_per_step'])
PY
done
