#!/bin/bash
# Round 5, GPU call 1: (1) baseline of the mapper iteration on the TRAINED map (kernel trace + SQ counters) before any kernel work of the round,
# (2) the pacing sweep of the in-system fused mapper on noisy sequences.   usage: bash tools/r05_call1.sh   (on the GPU box via gpurun)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05a
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
T="python $ROOT/bench.py --only trained --steps 50 --repeats 2"
$T > $OUT/trained_leg.json 2> $OUT/trained_leg.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_trained -o bench -- $T > $OUT/trained_leg_under_rocprof.json 2> $OUT/kt_trained.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/sq_trained -o p -- $T > /dev/null 2> $OUT/sq_trained.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2_trained -o p -- $T > /dev/null 2> $OUT/sq2_trained.err
cd $ROOT
timeout 700 python tools/pacing_sweep.py --out $OUT/fused_pacing_sweep.json --budget-s 520 > /dev/null 2> $OUT/pacing_sweep.log
find $OUT -name '*.csv' -size +20M -delete
tail -30 $OUT/pacing_sweep.log
tail -c 1500 $OUT/trained_leg.json
