#!/bin/bash
# Run ON THE GPU BOX through gpurun:  gpurun -- 'bash tools/profile_round.sh r01_v5'
# Produces under gpurun_out/<tag>/: bench.json (unprofiled run), kernel stats of the same command, and three PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters: separate runs, counters + kernel-trace only) that tools/pmc_traffic.py and
# tools/mfma_util.py summarise.  Copy kernel_stats.csv, pmc_traffic.{json,txt}, mfma_util.{json,txt}, bench.json to profiles/<tag>_*.
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-side --no-graph-replay > $out/bench_profiled.json 2> $out/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-side --no-graph-replay > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-side --no-graph-replay > /dev/null 2> $out/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-side --no-graph-replay > /dev/null 2> $out/pmc_sq.log
# bench.py --steps 3 --warmup 1 --windows 1 runs 1 warm-up + 2 (pre-pass that picks the dominant kernel) + 3 (per-kernel table pass) + 1 (warm-up in
# front of the timed steps) + 3 timed + 2 (full-table pass) = 12 step iterations, each launching every kernel of the merged tail once
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json 12 > $out/pmc_traffic.txt
python3 tools/mfma_util.py $out/pmc_sq $out/mfma_util.json 12 > $out/mfma_util.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
# raw traces are large; keep the summaries
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
cat $out/bench.json
