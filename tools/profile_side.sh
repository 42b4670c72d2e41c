#!/bin/bash
# Run ON THE GPU BOX:  gpurun -- 'bash tools/profile_side.sh r02_fm --workload fm --batch 1048576'
# A side benchmark (bench.py --workload fm|dcn|autoint): unprofiled run, rocprofv3 kernel stats of the same command, and
# the FETCH_SIZE / WRITE_SIZE passes (each its own run: --pmc with --kernel-trace only) summarised by tools/pmc_summary.py.
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --steps 10 --warmup 3 > $out/bench_profiled.json 2> $out/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc1 -- python3 bench.py "$@" --steps 3 --warmup 1 > /dev/null 2> $out/pmc1.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc2 -- python3 bench.py "$@" --steps 3 --warmup 1 > /dev/null 2> $out/pmc2.log
python3 tools/pmc_summary.py $out/pmc_summary.json $out/pmc1 $out/pmc2 > $out/pmc_summary.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
cat $out/bench.json; grep -A12 "^fm_\|^dcn_\|^attn_" $out/pmc_summary.txt | grep "^[a-z]\|hbm_bytes\|FETCH\|WRITE"
