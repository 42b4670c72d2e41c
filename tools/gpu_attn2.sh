#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/attn2
mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -m gpu -q -x -k "attn or autoint or AutoInt" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
for dl in 0 1; do
for L in 1 3; do
FIL_ATTN_DX_LDS=$dl python bench.py --workload autoint --precision f16_mfma --layers $L --steps 20 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('dx_lds=$dl L=$L', round(j['ms_per_step'],4), {k:v['avg_ms'] for k,v in j['kernels'].items()})
"
done
FIL_ATTN_DX_LDS=$dl python bench.py --workload autoint --precision f32 --layers 1 --steps 10 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('dx_lds=$dl f32 L=1', round(j['ms_per_step'],4), {k:v['avg_ms'] for k,v in j['kernels'].items()})
"
done
