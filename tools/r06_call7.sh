#!/bin/bash
# Round 6, GPU call 7: s_setprio on the LM kernel's waves (co-tenant with the mapper): headline step, tracker call profile.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06g
mkdir -p "$OUT"
export TMPDIR=/tmp
for rep in 1 2 3; do
for p in 0 3 1; do
  GSICP_TRACKER_WAVE_PRIO=$p python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --full-line --legs-file /tmp/l.json > $OUT/bench_prio${p}_$rep.json 2>> $OUT/err.log
  python -c "
import json; d=json.load(open('$OUT/bench_prio${p}_$rep.json')); print('prio $p rep $rep: ms_per_step', d['ms_per_step'], d['block_ms_per_step_p10_p50_p90'], 'align us (alone, eager)', d['stage_us_per_step'].get('gicp_align'))"
done
done
cd /tmp
for p in 0 3; do
  GSICP_TRACKER_WAVE_PRIO=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_prio$p -o bench -- python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs --legs-file /tmp/l0.json > $OUT/under_rocprof_prio$p.json 2> $OUT/kt_prio$p.err
done
cd $ROOT
find $OUT -name '*kernel_trace.csv' -delete
for p in 0 3; do echo == prio $p co-tenant; python - <<PY
import csv, re
for r in list(csv.DictReader(open("$OUT/kt_prio$p/bench_kernel_stats.csv")))[:10]:
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
timeout 300 env GSICP_TRACKER_WAVE_PRIO=3 python -m pytest tests/test_gicp_gpu.py -m gpu -x -q -k "lost_grid or sparse or pair or align" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1
