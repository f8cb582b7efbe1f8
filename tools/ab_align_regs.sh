#!/bin/bash
# A/B: the persistent LM kernel compiled under a register cap (amdgpu_waves_per_eu 2 / 3 / 4: 256 / 168 / 128 registers, spilling) against the
# product's 256 VGPRs + 114 AGPRs.  Variant libraries libgsicp_hip_wN.so (python tools/build_align_variants.py) are swapped in for libgsicp_hip.so on the (scratch) GPU box copy.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_align_regs
mkdir -p $OUT
cd $ROOT
cp gs_icp_slam_amd/libgsicp_hip.so /tmp/libgsicp_hip_product.so
for v in product w2 w3 w4; do
  if [ $v = product ]; then cp /tmp/libgsicp_hip_product.so gs_icp_slam_amd/libgsicp_hip.so; else cp gs_icp_slam_amd/libgsicp_hip_$v.so gs_icp_slam_amd/libgsicp_hip.so; fi
  python bench.py --no-cpu-baseline --no-legs > $OUT/full_$v.json 2>> $OUT/err.log
  python bench.py --only tracker --no-cpu-baseline --no-legs > $OUT/trk_$v.json 2>> $OUT/err.log
  python -c "
import json
f=json.loads(open('$OUT/full_$v.json').read().strip().splitlines()[-1]); t=json.loads(open('$OUT/trk_$v.json').read().strip().splitlines()[-1])
print('$v: step', f['ms_per_step'], f.get('block_ms_per_step'), 'tracker alone', t['ms_per_step'], 'pose err', f.get('pose_error_deg_mm'))"
done
cp /tmp/libgsicp_hip_product.so gs_icp_slam_amd/libgsicp_hip.so
