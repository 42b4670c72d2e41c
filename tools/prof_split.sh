#!/bin/bash
# Run ON THE GPU BOX: kernel stats of the CIN step (MODE=0 exact, default 2 = split-bf16: bench.py --cin-mode $MODE) -> gpurun_out/<tag>/kernel_stats.csv + bench line
set -u
tag=${1:-r06_split}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cin-mode ${MODE:-2} --no-candidate --no-side --no-cpu-baseline --no-graph-replay --windows 1 > $out/bench_profiled.json 2> $out/stats.log
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$out/kernel_stats.csv")))[:22]:
    print("%-70s %6s %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
PY
python3 -c "
import json; d=json.load(open('$out/bench_profiled.json')); print('ms_per_step', d['ms_per_step'])"
