#!/bin/bash
# Round-5 evidence in one GPU call: headline profile (kernel stats, PMC traffic, MFMA utilisation), the AutoInt config-5 profile,
# the batch sweep, and the side benchmarks.
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05_cin > /dev/null 2>&1
bash tools/profile_attn.sh r05_attn_f16_L3 f16_mfma 3 > /dev/null 2>&1
out=gpurun_out/r05_misc
mkdir -p $out
python tools/cin_batch_sweep.py > $out/cin_batch_sweep.txt 2>&1
python bench.py --force-collective --overlap on --no-cpu-baseline --no-side 2> $out/bench_fc.err | grep '^{' > $out/rccl_ws1_bench.json
python bench.py --force-collective --overlap off --no-cpu-baseline --no-side 2> /dev/null | grep '^{' > $out/rccl_ws1_no_overlap_bench.json
bash tools/gpu_model.sh xdeepfm > $out/xdeepfm_model.txt 2>&1
python bench.py --workload deepfm --graph > $out/deepfm_config2.json 2> /dev/null
python bench.py --workload dcn --graph --steps 200 --warmup 50 > $out/dcn_c3_bench.json 2> /dev/null
python bench.py --workload fm --graph --steps 200 --warmup 50 > $out/fm_c2_bench.json 2> /dev/null
python bench.py --workload fm --batch 1048576 > $out/fm_bench.json 2> /dev/null
python bench.py --workload autoint --precision f32 --layers 1 > $out/attn_f32_L1_bench.json 2> /dev/null
head -c 600 gpurun_out/r05_cin/bench.json; echo; cat gpurun_out/r05_cin/mfma_util.txt; cat gpurun_out/r05_cin/pmc_traffic.txt | grep -v "true>" | head -40; cat $out/cin_batch_sweep.txt
