#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3d
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_tail or benchmark_shape and cin" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -4 $out/pytest_cin.log
for st in 0 1; do
FIL_CIN_TAIL_SETTLE=$st python bench.py --no-cpu-baseline --no-side 2> $out/bench$st.err | grep '^{' > $out/bench$st.json
python -c "
import json
j=json.load(open('$out/bench$st.json')); print('settle=$st', j['value'], j['ms_per_step'], j.get('executed_frac'))
for k,v in j['kernels'].items():
    if 'executed_tflops' in v or v['avg_ms']>0.02: print('   ', k, v.get('avg_ms'), v.get('executed_tflops'))
"
done
for sp in 12 19 38 51; do
FIL_CIN_TAIL_SETTLE=1 FIL_CIN_TAIL_SPLITS=$sp python bench.py --no-cpu-baseline --no-side --steps 10 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('splits=$sp', j['ms_per_step'], j['kernels']['cin_bwd_dw_tail'])
"
done
