#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): captures the evidence that tools/collect_profiles.py turns into profiles/rNN_*.
# usage: bash tools/capture_profiles.sh r03      (outputs under gpurun_out/<tag>/)
# Counter passes are separate from the kernel-trace pass and never combined with other trace domains.
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the profiled command: the bench step without the extra legs (their kernels — other map sizes, torch's loss chain — would blur the averages)
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs"
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $ROOT/bench.py --res tum --no-cpu-baseline --no-legs > $OUT/bench_tum.json 2>> $OUT/bench.err
python $ROOT/bench.py --tracker pair --pair survey --no-cpu-baseline --no-legs > $OUT/bench_pair_survey.json 2>> $OUT/bench.err   # rounds 1-3's headline composite
python $ROOT/bench.py --tracker pair --pair basin --no-cpu-baseline --no-legs > $OUT/bench_pair_basin.json 2>> $OUT/bench.err
python $ROOT/bench.py --no-graph --no-cpu-baseline --no-legs > $OUT/bench_eager.json 2>> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $BENCH > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $BENCH > /dev/null 2> $OUT/write.err
# two passes of four counters (eight in one pass produced no counter file on this stack in round 4); tools/collect_profiles.py merges them
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq -o p -- $BENCH > /dev/null 2> $OUT/sq.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $BENCH > /dev/null 2> $OUT/sq2.err
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
(cd $ROOT && GSICP_ALIGN_TRACE=1 timeout 120 python tools/tracker_latency.py --map 300000 > $OUT/tracker_latency_map300k.txt 2>&1)
(cd $ROOT && timeout 200 python tools/tracker_vs_map.py 8280 100000 300000 1000000 3000000 > $OUT/tracker_vs_map.json 2>> $OUT/bench.err)
# the UNTOUCHED reference system on the drop-ins (synthetic sequences; one ray-cast sequence is shared through --cache): the 30-FPS-capped entry point
# (the map has to converge through the reference's own optimiser), the unlimited one with the drop-in call trace, and the TUM branch (TUM on-disk
# layout, tum.sh flags).  The 1500-frame unlimited run of round 3 is captured by tools/capture_reference_runs.sh.
cd $ROOT
C=/tmp/gsicp_cache
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 400 --timeout 500 --trace $OUT/trace_ref --log $OUT/reference_run_unlimit400.log > $OUT/reference_run_unlimit400.json 2> $OUT/reference_run.err
python tools/analyze_call_trace.py $OUT/trace_ref > $OUT/reference_call_trace.json 2>> $OUT/reference_run.err
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 300 --limit30 --timeout 500 > $OUT/reference_run_limit30_300.json 2>> $OUT/reference_run.err
# the same system with SURVEY 8(f)'s rows applied (oracle/make_refpy.py --fused): 400 frames unlimited with the call trace, 300 frames capped, 1500 frames unlimited (+ the untouched 1500)
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 400 --fused --timeout 500 --trace $OUT/trace_fused > $OUT/reference_run_fused_unlimit400.json 2>> $OUT/reference_run.err
python tools/analyze_call_trace.py $OUT/trace_fused > $OUT/reference_call_trace_fused.json 2>> $OUT/reference_run.err
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 300 --limit30 --fused --timeout 500 > $OUT/reference_run_fused_limit30_300.json 2>> $OUT/reference_run.err
timeout 900 python tools/run_reference_slam.py --cache $C --synthetic 1500 --timeout 700 > $OUT/reference_run_unlimit1500.json 2>> $OUT/reference_run.err
timeout 900 python tools/run_reference_slam.py --cache $C --synthetic 1500 --fused --timeout 700 > $OUT/reference_run_fused_unlimit1500.json 2>> $OUT/reference_run.err
timeout 400 python tools/run_reference_slam.py --synthetic 60 --shape tum --noise --limit30 --timeout 300 > $OUT/reference_run_tum_layout60.json 2>> $OUT/reference_run.err
# map quality: PSNR / SSIM / depth-L1 against mapper iterations (device-resident loop, ONE captured graph), and the scale-semantics coverage experiment
timeout 300 python tools/slam_demo.py 52 --iters 5 --prune-every 120 > $OUT/slam_demo.txt 2>&1
timeout 400 python tools/slam_demo.py 240 --iters 8 --eval-every 250 --post-iters 1500 --capacity 800000 --list-capacity 8388608 --no-asserts --json $OUT/map_quality_curve.json > $OUT/map_quality_curve.log 2>&1
timeout 120 python tools/scale_coverage.py --json $OUT/scale_coverage.json > /dev/null 2>&1
# counter calibration on kernels of known traffic
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o p -- python $ROOT/tools/pmc_calibration.py run > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o p -- python $ROOT/tools/pmc_calibration.py run > /dev/null 2>&1
cd $ROOT
python tools/pmc_calibration.py report $OUT/calib_fetch $OUT/calib_write > $OUT/pmc_calibration.json 2>/dev/null
# the contract command with N = 2 on this ONE GPU over gloo (launcher + both multi-GPU modes; functional rehearsal, not a scaling number)
GSICP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $OUT/bench_gpus2_gloo_one_gpu.json 2>> $OUT/bench.err
# keep only the small files (the merge-back limit is 64 MiB)
find $OUT -name '*.csv' -size +20M -delete
rm -rf $OUT/trace_ref $OUT/trace_fused $OUT/calib_fetch $OUT/calib_write
ls -la $OUT | head -60
