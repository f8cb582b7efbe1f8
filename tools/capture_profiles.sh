#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): captures the evidence that tools/collect_profiles.py turns into profiles/rNN_*.
# usage: bash tools/capture_profiles.sh r02      (outputs under gpurun_out/<tag>/)
# Counter passes are separate from the kernel-trace pass and never combined with other trace domains.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the profiled command: the bench step without the extra legs (their kernels — other map sizes, torch's loss chain — would blur the averages)
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs"
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $ROOT/bench.py --res tum --no-cpu-baseline > $OUT/bench_tum.json 2>> $OUT/bench.err
python $ROOT/bench.py --pair basin --no-cpu-baseline --no-legs > $OUT/bench_basin.json 2>> $OUT/bench.err
python $ROOT/bench.py --no-graph --no-cpu-baseline --no-legs > $OUT/bench_eager.json 2>> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
          --kernel-trace --output-format csv -d $OUT/sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
# each half ALONE under the kernel trace: per-kernel durations without the other half's co-tenancy (the iteration is the sum of these)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs > $OUT/bench_mapper_only_under_rocprof.json 2> $OUT/kt_mapper.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tracker -o bench -- python $ROOT/bench.py --only tracker --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs > $OUT/bench_tracker_only_under_rocprof.json 2> $OUT/kt_tracker.err
python $ROOT/bench.py --only mapper --no-cpu-baseline --no-legs > $OUT/bench_mapper_only.json 2>> $OUT/bench.err
python $ROOT/bench.py --only tracker --no-cpu-baseline --no-legs > $OUT/bench_tracker_only.json 2>> $OUT/bench.err
# the N > 1 iteration (tile movers + both RCCL collectives captured in the graph) on a 1-rank group: the exchange machinery without wire time
GSICP_BENCH_FORCE_COLLECTIVES=1 python $ROOT/bench.py --only mapper --no-cpu-baseline --no-legs > $OUT/bench_force_collectives.json 2>> $OUT/bench.err
python $ROOT/tools/rccl_graph_probe.py > $OUT/rccl_graph_probe.json 2>> $OUT/bench.err
# tracker: phase trace of the persistent LM kernel and the k-NN ring statistics on SURVEY 8(d)'s pair
(cd $ROOT && GSICP_ALIGN_TRACE=1 GSICP_KNN_STATS=1 timeout 120 python tools/tracker_latency.py --survey > $OUT/tracker_latency_survey.txt 2>&1)
# the UNTOUCHED reference system on the drop-ins (synthetic sequences in Replica's on-disk layout), with the drop-in call trace
cd $ROOT
timeout 600 python tools/run_reference_slam.py --synthetic 400 --timeout 500 --trace $OUT/trace_ref --log $OUT/reference_run_replica.log > $OUT/reference_run_replica.json 2> $OUT/reference_run.err
python tools/analyze_call_trace.py $OUT/trace_ref > $OUT/reference_call_trace.json 2>> $OUT/reference_run.err
timeout 400 python tools/run_reference_slam.py --synthetic 200 --shape tum --noise --timeout 300 > $OUT/reference_run_tum_shaped.json 2>> $OUT/reference_run.err
timeout 300 python tools/slam_demo.py 52 --iters 5 --prune-every 120 > $OUT/slam_demo.txt 2>&1
timeout 200 python tools/mfma_cov_experiment.py > $OUT/mfma_cov_experiment.json 2> /dev/null
# keep only the small files (the merge-back limit is 64 MiB)
find $OUT -name '*.csv' -size +20M -delete
rm -rf $OUT/trace_ref
ls -la $OUT | head -40
