#!/bin/bash
# Round 6, GPU call 1: the full GPU suite on the tree with the watchdog-safe capture, then the three RCCL-in-graph tests looped in fresh interpreters.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06a
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log | cut -c1-400
N=${1:-12}
ok=0
for i in $(seq 1 $N); do
  timeout 600 python -X faulthandler -m pytest -m gpu -x -q \
    "tests/test_bench_gpu.py::test_keyframes_mode_as_the_headline_on_a_one_rank_rccl_group_captured_in_the_graph" \
    "tests/test_sharded_gpu.py::test_captured_sharded_iteration_with_rccl_inside_equals_the_plain_graph" \
    "tests/test_raster_gpu.py::test_sharded_wrapper_collectives_on_rccl_world1" > $OUT/loop_$i.log 2>&1
  rc=$?
  [ $rc -eq 0 ] && ok=$((ok+1))
  echo "loop $i rc=$rc $(tail -1 $OUT/loop_$i.log)"
done
echo "rccl capture loop: $ok / $N green"
