#!/bin/bash
# Round 6's evidence capture (ON THE GPU BOX via gpurun; ~20 min): the contract line (default + the driver's command + every leg once), kernel traces of the
# step / each half alone / the trained-map leg, counter passes (separate from the traces and from each other), tracker evidence.
# usage: bash tools/capture_profiles_r06.sh [TAG]        then here: python tools/collect_profiles.py TAG r06
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/parity_report
cd /tmp
( time python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --legs-file $OUT/bench_legs_driver_cmd.json > $OUT/bench_driver_cmd.json 2> $OUT/bench.err ) 2> $OUT/bench_driver_cmd.time
( time python $ROOT/bench.py --legs-file $OUT/bench_legs.json > $OUT/bench.json 2>> $OUT/bench.err ) 2> $OUT/bench.time
python $ROOT/bench.py --all-legs --system-legs --legs-file $OUT/bench_all_legs.json > $OUT/bench_all_legs_line.json 2>> $OUT/bench.err
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs --legs-file /tmp/l0.json"
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/l1.json"
K="python $ROOT/bench.py --only tracker --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs --legs-file /tmp/l2.json"
T="python $ROOT/bench.py --only trained --steps 50 --repeats 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_mapper -o bench -- $M > $OUT/bench_mapper_only.json 2> $OUT/kt_mapper.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_tracker -o bench -- $K > $OUT/bench_tracker_only.json 2> $OUT/kt_tracker.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/bench_trained_leg.json 2> $OUT/kt_trained.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $M > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $M > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq -o p -- $M > /dev/null 2> $OUT/sq.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $M > /dev/null 2> $OUT/sq2.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq_trained -o p -- $T > /dev/null 2> $OUT/sq_trained.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2_trained -o p -- $T > /dev/null 2> $OUT/sq2_trained.err
$M > $OUT/bench_mapper_only_plain.json 2>> $OUT/bench.err
$K > $OUT/bench_tracker_only_plain.json 2>> $OUT/bench.err
cd $ROOT
(GSICP_ALIGN_TRACE=1 timeout 120 python tools/tracker_latency.py --map 300000 > $OUT/tracker_latency_map300k.txt 2>&1)
GSICP_PROFILES_DST=$OUT/profiles_out python tools/collect_profiles.py $TAG r06 > $OUT/collect.log 2>&1
tail -12 $OUT/collect.log
for f in bench_legs_driver_cmd bench_legs bench_all_legs bench_mapper_only_plain bench_tracker_only_plain; do cp $OUT/$f.json $OUT/profiles_out/r06_$f.json 2>/dev/null; done
cp $OUT/bench_all_legs_line.json $OUT/profiles_out/r06_bench_all_legs_line.json 2>/dev/null
cp $OUT/bench_driver_cmd.time $OUT/profiles_out/r06_bench_driver_cmd.time; cp $OUT/bench.time $OUT/profiles_out/r06_bench.time
rm -rf $OUT/kt $OUT/kt_mapper $OUT/kt_tracker $OUT/kt_trained $OUT/fetch $OUT/write $OUT/sq $OUT/sq2 $OUT/sq_trained $OUT/sq2_trained
rm -rf $ROOT/gpurun_out/reference_slam_*.log $ROOT/gpurun_out/trained_*.npz
du -sh $ROOT/gpurun_out
python - <<PY
import json
for n in ("bench_driver_cmd", "bench"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, len(json.dumps(d)), "chars:", d["value"], d["ms_per_step"], {k: d.get(k) for k in ("system_fps", "ate_cm", "psnr", "render_bwd_ms_per_iter", "mapper_only_ms_per_iter", "tracker_only_ms_per_frame", "mapper_iteration_ms_trained_map", "section_wall_s")})
    print("  roofline", d["roofline"]["frac"], d["roofline"]["kernel_us"], d["roofline"].get("traffic"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
d = json.load(open("$OUT/bench_all_legs.json")); print("all legs:", sorted(d["legs"].keys())); print({k: d.get(k) for k in ("system_fps","ate_cm","ate_cm_noisy","ate_cm_noisy_fused","fused_policy")})
print(open("$OUT/bench_driver_cmd.time").read())
PY
