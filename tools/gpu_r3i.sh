#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3i
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_cin and not promotion and not split and not benchmark_shape" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -4 $out/pytest_cin.log
for z in 1 0; do
FIL_CIN_DZS=$z python bench.py --no-cpu-baseline --no-side 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('dzs=$z', round(j['ms_per_step'],4), j['value'], 'dz_l1', k['cin_bwd_dz_l1'])
"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "benchmark_shape and cin" 2>&1 | tail -2
