#!/bin/bash
# kernel statistics of the whole-model side benchmarks (xDeepFM / DeepFM): what the step spends outside the interaction layer
cd $GRAFT_REPO_ROOT
w=${1:-xdeepfm}
out=gpurun_out/model_$w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --workload $w --graph --steps 20 > $out/bench.json 2> $out/bench.err
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --workload $w --steps 20 --warmup 5 > /dev/null 2> $out/stats.log
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (25 steps): %.3f ms" % (tot/25/1e6))
for r in rows[:45]:
    print("%-100s calls %4s avg %8.1f us  per-step %7.1f us" % (r["Name"].replace("void ","")[:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/25/1e3))
PY
