#!/bin/bash
# Round 5's evidence capture (ON THE GPU BOX via gpurun; ~15 min): the GPU suite, the contract line (default + the driver's command), kernel traces of the
# step / each half alone / the trained-map leg, counter passes (separate from the traces and from each other), tracker evidence.
# usage: bash tools/capture_profiles_r05.sh [TAG]        then here: python tools/collect_profiles.py TAG r05
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/parity_report
find $ROOT/gpurun_out -mindepth 1 -maxdepth 1 ! -name $TAG -exec rm -rf {} + 2>/dev/null    # the box starts without gpurun_out anyway; keep the result small
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1
tail -6 $OUT/pytest_gpu_final.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/pytest_gpu_final.log 2>&1
tail -1 $OUT/pytest_gpu_final.log
cd /tmp
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err
BENCH="python $ROOT/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs"
M="python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
K="python $ROOT/bench.py --only tracker --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs"
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
cd $ROOT
(GSICP_ALIGN_TRACE=1 timeout 120 python tools/tracker_latency.py --map 300000 > $OUT/tracker_latency_map300k.txt 2>&1)
# only gpurun_out/ comes back, and at most 64 MiB of it: summarise ON THE BOX into $OUT/profiles_out (what tools/collect_profiles.py would write into
# profiles/), then drop the raw traces and counter dumps (10-40 MB each)
GSICP_PROFILES_DST=$OUT/profiles_out python tools/collect_profiles.py $TAG r05 > $OUT/collect.log 2>&1
tail -12 $OUT/collect.log
rm -rf $OUT/kt $OUT/kt_mapper $OUT/kt_tracker $OUT/kt_trained $OUT/fetch $OUT/write $OUT/sq $OUT/sq2 $OUT/sq_trained $OUT/sq2_trained
rm -rf $ROOT/gpurun_out/reference_slam_*.log $ROOT/gpurun_out/trained_*.npz
du -sh $ROOT/gpurun_out
ls $OUT $OUT/profiles_out | head -70
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("bench", d["value"], d["ms_per_step"], {k: d.get(k) for k in ("system_fps", "ate_cm", "psnr", "ate_cm_noisy", "ate_cm_noisy_fused")})
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_us"], d["roofline"].get("traffic"))
lg = d["legs"]
print("trained", lg["mapper_trained_map"].get("ms_per_iteration"), "tum", lg.get("step_tum", {}).get("ms_per_step"), "mapper_only", lg["mapper_only"]["ms_per_iteration"])
d = json.load(open("$OUT/bench_driver_cmd.json")); print("driver cmd", d["value"], d["ms_per_step"])
PY
