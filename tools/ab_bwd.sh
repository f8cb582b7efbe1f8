#!/bin/bash
# Rasteriser / graph / sharded GPU tests, then the mapper-only bench under rocprofv3 (blend kernel averages) and alone (iteration time).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_bwd_${1:-x}
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_graph_gpu.py tests/test_sharded_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/tests.log | tail -3
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python $ROOT/bench.py --only mapper --steps 50 --warmup 5 --repeats 2 --no-cpu-baseline --no-legs > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
grep -h "blend_backward\|blend_forward\|tile_sort" $OUT/kt/bench_kernel_stats.csv | cut -c1-220
python $ROOT/bench.py --only mapper --no-cpu-baseline --no-legs > $OUT/bench_alone.json 2>> $OUT/kt.err
python -c "import json; d=json.loads(open('$OUT/bench_alone.json').read().strip().splitlines()[-1]); print('mapper ms_per_step', d['ms_per_step'])"
python $ROOT/bench.py --no-cpu-baseline --no-legs > $OUT/bench_full.json 2>> $OUT/kt.err
python -c "import json; d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]); print('full ms_per_step', d['ms_per_step'], d['value'])"
rm -f $OUT/kt/*kernel_trace.csv
