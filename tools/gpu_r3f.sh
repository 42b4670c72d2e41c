#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3f
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py tests/test_properties.py -m gpu -q -x -k "cin or CIN or xdeepfm or XDeepFM or zoo" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -4 $out/pytest_cin.log
python bench.py --no-cpu-baseline --no-side 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print(round(j['ms_per_step'],4), j['value'])
for n,v in k.items(): print('  ', n, v['avg_ms'], v.get('executed_tflops'))
"
bash tools/gpu_sq.sh sq3 2>&1 | grep -v "dw3b\|true>\|split_g\|pack_wb\|pack_wzb\|last_bwd2\|scale_rows\|fill_rows\|slice_sum\|wsum_wsn\|dw3_kernel<1, true" | tail -34
