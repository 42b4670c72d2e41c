#!/bin/bash
# Round-3 evidence in one GPU call: headline profile (kernel stats, PMC traffic, MFMA utilisation), the AutoInt config-5 profile,
# the batch sweep, the forced one-rank collective (timed + kernel trace) and the whole-model kernel statistics.
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03_cin > /dev/null 2>&1
bash tools/profile_attn.sh r03_attn_f16_L3 f16_mfma 3 > /dev/null 2>&1
out=gpurun_out/r03_misc
mkdir -p $out
python tools/cin_batch_sweep.py > $out/cin_batch_sweep.txt 2>&1
FIL_CIN_KSPLIT=0 python tools/cin_batch_sweep.py 2>&1 | head -3 | sed 's/^/no reduction split: /' >> $out/cin_batch_sweep.txt
python bench.py --force-collective --no-cpu-baseline --no-side 2> $out/bench_fc.err | grep '^{' > $out/rccl_ws1_bench.json
python bench.py --force-collective --no-overlap --no-cpu-baseline --no-side 2> /dev/null | grep '^{' > $out/rccl_ws1_no_overlap_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rccl -- python3 bench.py --force-collective --no-cpu-baseline --no-side --steps 6 --warmup 2 > /dev/null 2> $out/rccl.log
python tools/trace_order.py $out/rccl 3 > $out/rccl_ws1_trace_order.txt 2>&1
find $out -name "*kernel_trace.csv" -delete
bash tools/gpu_model.sh xdeepfm > $out/xdeepfm_model.txt 2>&1
python bench.py --workload deepfm --graph > $out/deepfm_config2.json 2> /dev/null
python bench.py --workload dcn --graph > $out/dcn_c3_bench.json 2> /dev/null
python bench.py --workload fm --batch 1048576 > $out/fm_bench.json 2> /dev/null
python bench.py --workload autoint --precision f32 --layers 1 > $out/attn_f32_L1_bench.json 2> /dev/null
head -c 400 gpurun_out/r03_cin/bench.json; echo; cat gpurun_out/r03_cin/mfma_util.txt; cat gpurun_out/r03_cin/pmc_traffic.txt | grep -v "true>" | head -30; cat $out/cin_batch_sweep.txt; head -30 $out/rccl_ws1_trace_order.txt; cat gpurun_out/r03_attn_f16_L3/pmc_summary.txt | head -40
