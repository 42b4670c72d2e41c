#!/bin/bash
# diagnostic: SQ counter pass + kernel stats of the default bench (no PMC traffic passes)
cd $GRAFT_REPO_ROOT
tag=${1:-sq}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-side --steps 10 > $out/bench_profiled.json 2> $out/stats.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > /dev/null 2> $out/pmc_sq.log
python3 tools/mfma_util.py $out/pmc_sq $out/mfma_util.json 6 > $out/mfma_util.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
cat $out/mfma_util.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:40]:
    print("%-90s calls %4s avg %9.1f us  %5s%%" % (r["Name"].replace("void fil::","").split("(")[0][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
