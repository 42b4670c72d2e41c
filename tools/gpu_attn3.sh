#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 1 3; do
python bench.py --workload autoint --precision f16_mfma --layers $L --steps 20 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('L=$L', round(j['ms_per_step'],4), {k:v['avg_ms'] for k,v in j['kernels'].items()})
"
done
