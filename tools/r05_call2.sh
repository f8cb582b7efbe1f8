#!/bin/bash
# Round 5, GPU call 2: the new per-Gaussian backward pass (tests + timings on the S-map and the trained map) and the confirmation runs of the pacing sweep.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py -x -q > $OUT/pytest_raster.log 2>&1
tail -15 $OUT/pytest_raster.log
cd /tmp
T="python $ROOT/bench.py --only trained --steps 50 --repeats 2"
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/trained_leg.json 2> $OUT/kt_trained.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/mapper_only.json 2> $OUT/kt_mapper.err
GSICP_PREBWD_LEGACY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper_legacy -o bench -- $M > $OUT/mapper_only_legacy.json 2> $OUT/kt_mapper_legacy.err
cd $ROOT
timeout 600 python tools/pacing_sweep.py --plan v2 --out $OUT/fused_pacing_sweep_v2.json --budget-s 420 > /dev/null 2> $OUT/pacing_sweep_v2.log
find $OUT -name '*.csv' -size +20M -delete
cat $OUT/pacing_sweep_v2.log
for d in kt_trained kt_mapper kt_mapper_legacy; do echo == $d; head -12 $OUT/$d/bench_kernel_stats.csv | cut -c1-60,100-400 | awk -F'"' '{print substr($2,1,70), $0}' | awk '{print $NF}' > /dev/null; python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/$d/bench_kernel_stats.csv")))[:14]:
    import re
    m = re.search(r'(\w+_kernel)', r["Name"])
    print("%-34s calls %5s avg_us %9.2f" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
