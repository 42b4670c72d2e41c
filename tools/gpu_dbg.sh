#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "FIL_DZS_DBG=0" "FIL_DZS_DBG=1" "FIL_DZS_DBG=0 FIL_CIN_MB=1" "FIL_DZS_DBG=1 FIL_CIN_MB=1" "FIL_CIN_DZS=0 FIL_CIN_MB=1"; do
env $e python bench.py --no-cpu-baseline --no-side --steps 10 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$e', round(j['ms_per_step'],4), 'dz_l1', k['cin_bwd_dz_l1']['avg_ms'], 'fwd_l1', k['cin_fwd_l1']['avg_ms'])
"
done
