#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-side --cin-mode 32 --steps 1 --warmup 0 > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-side 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print(round(j['ms_per_step'],4), j['value'])
for n,v in k.items():
    if 'executed_tflops' in v: print('  ', n, v['avg_ms'], v.get('executed_tflops'))
"
