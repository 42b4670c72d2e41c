#!/bin/bash
# round 3, first GPU pass: the whole -m gpu suite, the default bench line, the forced one-rank collective (timed + traced)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3a
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --force-collective --no-cpu-baseline > $out/bench_fc.json 2> $out/bench_fc.err
python bench.py --force-collective --no-overlap --no-cpu-baseline > $out/bench_fc_noov.json 2> $out/bench_fc_noov.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rccl -- python3 bench.py --force-collective --no-cpu-baseline --steps 6 --warmup 2 > $out/rccl_bench.json 2> $out/rccl.log
python tools/trace_order.py $out/rccl 3 > $out/rccl_trace_order.txt 2>&1
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/rccl_kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
head -c 600 $out/bench.json; echo
python -c "
import json
for f in ('bench_fc','bench_fc_noov'):
    try:
        print(f, json.dumps(json.load(open('$out/'+f+'.json'))['rccl']))
    except Exception as e: print(f, 'ERR', e)
"
