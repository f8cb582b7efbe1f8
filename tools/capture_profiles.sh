#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): captures the evidence that tools/collect_profiles.py turns into profiles/rNN_*.
# usage: bash tools/capture_profiles.sh r01      (outputs under gpurun_out/<tag>/)
# Counter passes are separate from the kernel-trace pass and never combined with other trace domains.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $ROOT/bench.py --res tum --no-cpu-baseline > $OUT/bench_tum.json 2>> $OUT/bench.err
python $ROOT/bench.py --no-graph --no-cpu-baseline > $OUT/bench_eager.json 2>> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
          --kernel-trace --output-format csv -d $OUT/sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
# keep only the small CSVs (the merge-back limit is 64 MiB)
find $OUT -name '*.csv' -size +20M -delete
ls -la $OUT $OUT/kt 2>/dev/null | head -40
