#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): re-captures only the bench lines + the co-tenant kernel trace of tools/capture_profiles.sh (after a change
# that affects the timed loop but not the kernels).  usage: bash tools/capture_bench_lines.sh r04
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs"
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $ROOT/bench.py --no-legs --no-cpu-baseline > $OUT/bench_second_run.json 2>> $OUT/bench.err
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err     # the driver's command (round 3's BENCH record)
python $ROOT/bench.py --res tum --no-cpu-baseline --no-legs > $OUT/bench_tum.json 2>> $OUT/bench.err
python $ROOT/bench.py --tracker pair --pair survey --no-cpu-baseline --no-legs > $OUT/bench_pair_survey.json 2>> $OUT/bench.err
python $ROOT/bench.py --tracker pair --pair basin --no-cpu-baseline --no-legs > $OUT/bench_pair_basin.json 2>> $OUT/bench.err
python $ROOT/bench.py --no-graph --no-cpu-baseline --no-legs > $OUT/bench_eager.json 2>> $OUT/bench.err
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $ROOT/bench.py --only mapper --no-cpu-baseline --no-legs > $OUT/bench_mapper_only.json 2>> $OUT/bench.err
python $ROOT/bench.py --only tracker --no-cpu-baseline --no-legs > $OUT/bench_tracker_only.json 2>> $OUT/bench.err
cd $ROOT
GSICP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-reference-leg > $OUT/bench_gpus2_gloo_one_gpu.json 2>> $OUT/bench.err
find $OUT -name '*.csv' -size +20M -delete
ls -la $OUT | head -40
