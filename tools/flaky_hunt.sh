#!/bin/bash
# Repeats the in-process GPU test files N times, each in a fresh interpreter, and keeps every log (a crash of the interpreter shows as rc != 0).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/flaky
mkdir -p $OUT
cd $ROOT
N=${1:-4}
for i in $(seq 1 $N); do
  timeout 300 python -X faulthandler -m pytest tests/test_sharded_gpu.py tests/test_slam_loop_gpu.py tests/test_store_gpu.py tests/test_handoff_gpu.py tests/test_graph_gpu.py tests/test_knn_gpu.py tests/test_mapper_ops_gpu.py -m gpu -x -v > $OUT/run_$i.log 2>&1
  echo "run $i rc=$? $(grep -c PASSED $OUT/run_$i.log) passed"
done
