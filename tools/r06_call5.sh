#!/bin/bash
# Round 6, GPU call 5: k-NN grid cell-size sweep (exact for any value) on the tracker's steady-state frame; store growth test.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06e
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_store_gpu.py -m gpu -x -q > $OUT/pytest_store.log 2>&1
grep -E "passed|failed" $OUT/pytest_store.log | tail -1
for rep in 1 2; do
for h in 1.0 1.25 1.5 1.75 2.0 2.25 2.5; do
  GSICP_KNN_H=$h python bench.py --only tracker --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-legs --min-seconds 0.5 --full-line --legs-file /tmp/l.json > $OUT/trk_h${h}_$rep.json 2>> $OUT/err.log
  python -c "
import json; d=json.load(open('$OUT/trk_h${h}_$rep.json')); s=d['stage_us_per_step']; print('h $h rep $rep: ms_per_frame', d['ms_per_step'], 'knn_cov us', s.get('gicp_knn_cov'), 'align', s.get('gicp_align'), 'exact_nn', s.get('gicp_exact_nn'))"
done
done
tail -2 $OUT/err.log
