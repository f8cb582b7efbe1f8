#!/bin/bash
# Round 5, GPU call 6: run-to-run spread of the contract line under the driver's command (20-step blocks) and the default (100-step blocks), legs off.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05g
mkdir -p "$OUT"
cd /tmp
for i in 1 2 3 4; do
  python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $OUT/driver_cmd_$i.json 2>> $OUT/err.log
done
for i in 1 2; do
  python3 $ROOT/bench.py --no-legs --no-cpu-baseline > $OUT/default_$i.json 2>> $OUT/err.log
done
GSICP_PREBWD_LEGACY=1 GSICP_EMIT_WALK=1 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $OUT/driver_cmd_legacy_kernels.json 2>> $OUT/err.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["block_ms_per_step_p10_p50_p90"], d["repeats"])
PY
