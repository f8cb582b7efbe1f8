#!/bin/bash
# What does one more kernel boundary in the tracker's frame cost, alone and next to the mapper?  N empty launches in front of the LM kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_boundary
mkdir -p $OUT
cd $ROOT
for n in 0 8 16; do
  GSICP_TRACKER_EXTRA_LAUNCHES=$n python bench.py --no-cpu-baseline --no-legs > $OUT/full_$n.json 2>> $OUT/err.log
  GSICP_TRACKER_EXTRA_LAUNCHES=$n python bench.py --only tracker --no-cpu-baseline --no-legs > $OUT/trk_$n.json 2>> $OUT/err.log
  python -c "
import json
f=json.loads(open('$OUT/full_$n.json').read().strip().splitlines()[-1]); t=json.loads(open('$OUT/trk_$n.json').read().strip().splitlines()[-1])
print('extra launches $n: step', f['ms_per_step'], 'tracker alone', t['ms_per_step'])"
done
