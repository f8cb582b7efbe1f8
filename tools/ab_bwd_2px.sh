#!/bin/bash
# A/B of the blend backward: four waves x one pixel per lane (default) against two waves x two pixels per lane (GSICP_BWD_2PX=1).
# Runs the rasteriser's backward parity tests under the variant, then the mapper-only bench under rocprofv3 for both.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_bwd_2px
mkdir -p $OUT
cd $ROOT
GSICP_BWD_2PX=1 timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $OUT/tests_2px.log 2>&1
tail -3 $OUT/tests_2px.log
cd /tmp
for v in 0 1; do
  GSICP_BWD_2PX=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -o bench -- python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs > $OUT/bench_$v.json 2> $OUT/kt_$v.err
  grep -h "blend_backward\|blend_forward" $OUT/kt_$v/bench_kernel_stats.csv | cut -c1-200
  GSICP_BWD_2PX=$v python $ROOT/bench.py --only mapper --no-cpu-baseline --no-legs > $OUT/bench_alone_$v.json 2>> $OUT/kt_$v.err
  python -c "import json,sys; d=json.loads(open('$OUT/bench_alone_$v.json').read().strip().splitlines()[-1]); print('variant $v ms_per_step', d['ms_per_step'])"
  rm -f $OUT/kt_$v/*kernel_trace.csv
done
