#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3h
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_cin and not promotion and not split" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -4 $out/pytest_cin.log
python bench.py --no-cpu-baseline --no-side 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print(round(j['ms_per_step'],4), j['value'])
for n,v in k.items(): print('  ', n, v['avg_ms'], v.get('executed_tflops'))
print('split', j['candidate_split_bf16']['ms_per_step'])
"
