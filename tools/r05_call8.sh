#!/bin/bash
# Round 5, GPU call 8: the loss kernels without their depth slice (grid z = 3): tests + A/B trace.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05i
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mapper_ops_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py tests/test_slam_loop_gpu.py -q > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log | cut -c1-300
cd /tmp
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/mapper_only.json 2> $OUT/kt_mapper.err
GSICP_LOSS_DEPTH_IN_CH0=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_depth_slice -o bench -- $M > $OUT/mapper_only_depth_slice.json 2> $OUT/kt_mapper_d.err
$M > $OUT/mapper_only_plain.json 2>> $OUT/kt_mapper.err
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for d in kt_mapper kt_mapper_depth_slice; do echo == $d; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/$d/bench_kernel_stats.csv")))[:8]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
python -c "
import json
d=json.load(open('$OUT/mapper_only_plain.json')); print('mapper_only', d['ms_per_step'], d.get('value'))
"
