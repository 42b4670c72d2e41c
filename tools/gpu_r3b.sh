#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3b
mkdir -p $out
python tools/diag_attn_f16.py > $out/diag_attn.txt 2>&1
python -m pytest tests -m gpu -q --deselect "tests/test_golden.py::test_gpu_autoint_stack_kink_free" > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -30 $out/pytest.log
python bench.py --force-collective --no-cpu-baseline 2> $out/bench_fc.err | grep '^{' > $out/bench_fc.json
python bench.py --force-collective --no-overlap --no-cpu-baseline 2> $out/bench_fc_noov.err | grep '^{' > $out/bench_fc_noov.json
cat $out/diag_attn.txt
python -c "
import json
for f in ('bench_fc','bench_fc_noov'):
    j=json.load(open('$out/'+f+'.json')); print(f, j['ms_per_step'], json.dumps(j['rccl']))
"
