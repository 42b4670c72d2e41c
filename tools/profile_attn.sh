#!/bin/bash
# Run ON THE GPU BOX:  gpurun -- 'bash tools/profile_attn.sh r02_attn f16_mfma 3'
# AutoInt side benchmark (BASELINE config 5): unprofiled run, rocprofv3 kernel stats of the same command, and PMC passes
# (each its own run: --pmc with --kernel-trace only).  Summaries land in gpurun_out/<tag>/; copy them to profiles/.
set -u
tag=${1:-attn}; prec=${2:-f16_mfma}; L=${3:-3}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--workload autoint --precision $prec --layers $L"
python3 bench.py $args > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $args --steps 10 --warmup 3 > $out/bench_profiled.json 2> $out/stats.log
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/pmc$i -- python3 bench.py $args --steps 3 --warmup 1 > /dev/null 2> $out/pmc$i.log
done
python3 tools/pmc_summary.py $out/pmc_summary.json $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 > $out/pmc_summary.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
cat $out/bench.json; cat $out/pmc_summary.txt
