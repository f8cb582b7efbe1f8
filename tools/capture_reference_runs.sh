#!/bin/bash
# Round-3 evidence runs of the reference's UNMODIFIED two-process system on the drop-ins: one ray-cast Replica-shaped
# sequence serves the 1500-frame unlimit run, the 400-frame unlimit run and the 400-frame run of the 30-FPS-capped entry point
# [REF mp_Tracker.py:323-324]; plus the TUM branch (TUM on-disk layout, tum.sh flags).   usage: bash tools/capture_reference_runs.sh OUTDIR
OUT=${1:-gpurun_out/refruns}
mkdir -p "$OUT"
C=/tmp/gsicp_cache
run() { tag=$1; shift; python tools/run_reference_slam.py --cache $C --log "$OUT/$tag.log" "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.err"; tail -c 900 "$OUT/$tag.json"; echo; }
run unlimit1500 --synthetic 1500 --timeout 800
run unlimit400 --synthetic 400 --timeout 400 --trace "$OUT/trace400"
run limit30_400 --synthetic 400 --limit30 --timeout 400
run tum_layout200 --synthetic 200 --shape tum --noise --timeout 400
python tools/analyze_call_trace.py "$OUT/trace400" > "$OUT/trace400.json" 2>/dev/null
