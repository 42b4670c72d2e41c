#!/bin/bash
# Run ON THE GPU BOX: kernel stats of a whole-model workload -> gpurun_out/<tag>/kernel_stats.csv ; usage: prof_model.sh <tag> <workload>
set -u
tag=${1:-r06_model}; wl=${2:-xdeepfm}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --workload $wl --steps 20 --warmup 5 > $out/bench_profiled.json 2> $out/stats.log
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/kernel_stats.csv")))
tot = 0.0
for r in rows[:45]:
    per_step = float(r["TotalDurationNs"]) / 25.0 / 1000.0
    tot += per_step
    print("%-90s %6s %9.1f us  %8.1f us/step" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1000.0, per_step))
print("sum of listed, us/step:", tot)
PY
