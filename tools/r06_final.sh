#!/bin/bash
# Round 6, END-OF-ROUND PROTOCOL: on the exact tree that is handed in — (a) the full GPU suite with -x, (b) the driver's bench command, (c) the three
# RCCL-in-hipGraph tests looped in fresh interpreters, (d) smoke().  usage: bash tools/r06_final.sh <git hash of the tree> [loops]
set -u
HASH=${1:-unknown}
LOOPS=${2:-30}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06_final
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/parity_report
echo "# tree = git $HASH ; python -m pytest tests -m gpu -x -q ; $(date -u +%FT%TZ)" > $OUT/pytest_gpu_final.log
( time timeout 2400 python -m pytest tests -m gpu -x -q ) >> $OUT/pytest_gpu_final.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/pytest_gpu_final.log 2>&1
tail -1 $OUT/pytest_gpu_final.log
cd /tmp
echo "# tree = git $HASH ; python3 bench.py --gpus 1 --steps 20 --warmup 5" > $OUT/bench_driver_cmd.time
( time python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --legs-file $OUT/bench_legs_driver_cmd.json > $OUT/bench_driver_cmd.json 2> $OUT/bench.err ) 2>> $OUT/bench_driver_cmd.time
tail -4 $OUT/bench_driver_cmd.time
wc -c $OUT/bench_driver_cmd.json
cd $ROOT
echo "# tree = git $HASH ; the three RCCL-in-hipGraph tests, $LOOPS fresh interpreters" > $OUT/rccl_capture_loop.txt
ok=0
for i in $(seq 1 $LOOPS); do
  timeout 600 python -X faulthandler -m pytest -m gpu -x -q \
    "tests/test_bench_gpu.py::test_keyframes_mode_as_the_headline_on_a_one_rank_rccl_group_captured_in_the_graph" \
    "tests/test_sharded_gpu.py::test_captured_sharded_iteration_with_rccl_inside_equals_the_plain_graph" \
    "tests/test_raster_gpu.py::test_sharded_wrapper_collectives_on_rccl_world1" > $OUT/loop.log 2>&1
  rc=$?
  [ $rc -eq 0 ] && ok=$((ok+1))
  echo "loop $i rc=$rc $(grep -E 'passed|failed' $OUT/loop.log | tail -1)" >> $OUT/rccl_capture_loop.txt
  [ $rc -ne 0 ] && cp $OUT/loop.log $OUT/loop_failed_$i.log
done
echo "rccl capture loop: $ok / $LOOPS green" | tee -a $OUT/rccl_capture_loop.txt
python tools/collect_profiles.py r06_final r06 > /dev/null 2>&1 || true
mkdir -p $OUT/profiles_out
GSICP_PROFILES_DST=$OUT/profiles_out python tools/collect_profiles.py r06_final r06 > $OUT/collect.log 2>&1
ls $OUT/profiles_out
python - <<PY
import json
d = json.load(open("$OUT/bench_driver_cmd.json"))
print(len(json.dumps(d)), "chars", d["value"], d["ms_per_step"], {k: d.get(k) for k in ("system_fps", "ate_cm", "psnr", "step_tum_ms", "mapper_only_ms_per_iter", "tracker_only_ms_per_frame", "mapper_iteration_ms_trained_map", "section_wall_s")})
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_us"], d["roofline"].get("traffic"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
rm -rf $ROOT/gpurun_out/reference_slam_*.log $ROOT/gpurun_out/trained_*.npz
