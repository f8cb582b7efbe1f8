#!/bin/bash
# Round 5, GPU call 7: the default policy on LONGER runs: 1 500 noise-free frames, 600 noisy frames with fast motion (untouched and fused).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05h
mkdir -p "$OUT"
C=/tmp/gsicp_cache
timeout 900 python tools/run_reference_slam.py --cache $C --synthetic 1500 --fused --timeout 700 > $OUT/reference_run_fused_unlimit1500.json 2> $OUT/err.log
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 600 --noise --speed 2 --jitter 0.003 --timeout 500 > $OUT/reference_run_noisy600.json 2>> $OUT/err.log
timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 600 --noise --speed 2 --jitter 0.003 --fused --timeout 500 > $OUT/reference_run_noisy600_fused.json 2>> $OUT/err.log
GSICP_FUSED_POLICY=budget timeout 600 python tools/run_reference_slam.py --cache $C --synthetic 600 --noise --speed 2 --jitter 0.003 --fused --timeout 500 > $OUT/reference_run_noisy600_fused_budget.json 2>> $OUT/err.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    fm = d.get("fused_mapper") or {}
    print(f.split("/")[-1], {k: d.get(k) for k in ("system_fps", "ate_rmse_cm", "ate_true_rmse_cm", "ate_max_cm", "psnr", "ssim", "wall_s")}, fm.get("iterations"), fm.get("gpu_median_ms_per_iteration"), fm.get("policy"), fm.get("gaussians"))
PY
tail -3 $OUT/err.log
