#!/bin/bash
# Round 6, GPU call 11: (a) the fused in-system run with a map capacity small enough to force GaussianStore.grow() (and re-captures) inside the reference's two-process
# system, against the same run with room from the start; (b) a 1 500-frame fused run on the final kernels; (c) run-to-run spread of the driver's command.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06k
mkdir -p "$OUT"
export TMPDIR=/tmp
C=/tmp/gsicp_synth_cache
timeout 400 python tools/run_reference_slam.py --cache $C --synthetic 300 --fused --timeout 300 --log $OUT/fused_roomy.log > $OUT/fused_roomy.json 2>> $OUT/err.log
GSICP_FUSED_CAPACITY=40000 timeout 400 python tools/run_reference_slam.py --cache $C --synthetic 300 --fused --timeout 300 --log $OUT/fused_grow.log > $OUT/fused_grow.json 2>> $OUT/err.log
grep -c "growing to" $OUT/fused_grow.log; grep "growing to" $OUT/fused_grow.log | head -5
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 1500 --fused --timeout 500 > $OUT/fused_1500.json 2>> $OUT/err.log
python - <<PY
import json
for n in ("fused_roomy", "fused_grow", "fused_1500"):
    try:
        d = json.loads([l for l in open("$OUT/%s.json" % n).read().splitlines() if l.startswith("{")][-1])
        fm = d.get("fused_mapper") or {}
        print(n, {k: d.get(k) for k in ("status", "system_fps", "ate_rmse_cm", "ate_true_rmse_cm", "psnr", "ssim", "frames")}, {k: fm.get(k) for k in ("iterations", "graph_captures", "gaussians", "gpu_median_ms_per_iteration", "policy")})
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp
for i in 1 2 3 4 5; do
  python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --legs-file /tmp/l.json > $OUT/driver_cmd_$i.json 2>> $OUT/err.log
  python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print('run $i', d['ms_per_step'], d['block_ms_per_step_p10_p50_p90'])"
done
tail -3 $OUT/err.log
