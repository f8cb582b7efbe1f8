#!/bin/bash
# A/B: the persistent LM kernel on fewer workgroups (GSICP_ALIGN_WG), headline step and tracker-only frame.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_align_wg
mkdir -p $OUT
cd $ROOT
for wg in 0 24 16 8; do
  GSICP_ALIGN_WG=$wg python bench.py --no-cpu-baseline --no-legs > $OUT/full_$wg.json 2>> $OUT/err.log
  GSICP_ALIGN_WG=$wg python bench.py --only tracker --no-cpu-baseline --no-legs > $OUT/trk_$wg.json 2>> $OUT/err.log
  python -c "
import json
f=json.loads(open('$OUT/full_$wg.json').read().strip().splitlines()[-1]); t=json.loads(open('$OUT/trk_$wg.json').read().strip().splitlines()[-1])
print('wg cap $wg: step', f['ms_per_step'], f.get('block_ms_per_step'), 'tracker alone', t['ms_per_step'])"
done
