#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3c
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cin" > $out/pytest_cin.log 2>&1
echo "pytest rc=$?" >> $out/pytest_cin.log
tail -25 $out/pytest_cin.log
python bench.py --no-cpu-baseline 2> $out/bench.err | grep '^{' > $out/bench.json
python -c "
import json
j=json.load(open('$out/bench.json')); print(j['value'], j['ms_per_step']); print(json.dumps(j.get('kernels'), indent=0))
"
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -s -k "attn_at_the_benchmark or autoint_stack" > $out/pytest_attn.log 2>&1
grep -E "c5 at B|passed|failed|Error" $out/pytest_attn.log | cut -c1-600
