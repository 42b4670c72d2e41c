#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3e
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_tail" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -3 $out/pytest_cin.log
run() {
  env "$@" python bench.py --no-cpu-baseline --no-side --steps 10 2> /dev/null | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels']
print('$*', round(j['ms_per_step'],4), 'fwd', k['cin_fwd_tail']['avg_ms'], 'dw', k['cin_bwd_dw_tail']['avg_ms'], 'dz', k['cin_bwd_dz_tail']['avg_ms'])
"
}
run FIL_CIN_TAIL_SETTLE=0
run FIL_CIN_TAIL_SETTLE=1
run FIL_CIN_TAIL_SETTLE=1 FIL_CIN_TAIL_DZ_MODE=1
run FIL_CIN_TAIL_SETTLE=1 FIL_CIN_TAIL_DZ_MODE=2
run FIL_CIN_TAIL_SETTLE=1 FIL_CIN_TAIL_DZ_MODE=3
run FIL_CIN_TAIL_SETTLE=1 FIL_CIN_TAIL_SPLITS=38
run FIL_CIN_TAIL_SETTLE=0 FIL_CIN_TAIL_SPLITS=38
