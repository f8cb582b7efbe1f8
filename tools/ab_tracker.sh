#!/bin/bash
# Tracker-side change check: every tracker / hand-off / loop test, then the headline step, the tracker alone and the per-call host profile.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_tracker_${1:-x}
mkdir -p $OUT
cd $ROOT
timeout 500 python -X faulthandler -m pytest tests/test_gicp_gpu.py tests/test_handoff_gpu.py tests/test_slam_loop_gpu.py tests/test_pybind_module.py tests/test_frontend.py tests/test_hostcode_pinned.py -m gpu -x -q > $OUT/tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/tests.log | tail -4
GSICP_BENCH_PMC=0 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d.get('block_ms_per_step'), d['value'])
L=d['legs']; print('tracker alone', L['tracker_only_survey']['ms_per_frame'], 'basin', L['tracker_only_basin']['ms_per_frame'], 'mapper', L['mapper_only']['ms_per_iteration'], 'lockstep', L.get('lockstep_step',{}).get('ms_per_step'))
print(json.dumps(L['tracker_call_profile_us']))"
