#!/bin/bash
# Round-6 evidence in one GPU call: headline profile (kernel stats, PMC traffic, MFMA utilisation), the split-bf16 candidate's kernel
# stats + SQ counters, the AutoInt config-5 profile, the c2 FM / c3 DCN side profiles at their BASELINE batch sizes (kernel stats + PMC
# traffic: bench.py's roofline.traffic lookups), the batch sweep, the one-rank RCCL plumbing numbers and the whole-model steps.
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06_cin > /dev/null 2>&1
bash tools/prof_split.sh r06_cin_bf16x3 > /dev/null 2>&1
bash tools/pmc_split.sh r06_cin_bf16x3_sq > /dev/null 2>&1
bash tools/profile_attn.sh r06_attn_f16_L3 f16_mfma 3 > /dev/null 2>&1
bash tools/profile_side.sh r06_fm_c2 --workload fm > /dev/null 2>&1
bash tools/profile_side.sh r06_dcn_c3 --workload dcn > /dev/null 2>&1
out=gpurun_out/r06_misc
mkdir -p $out
python tools/cin_batch_sweep.py > $out/cin_batch_sweep.txt 2>&1
python bench.py --force-collective --overlap on --no-cpu-baseline --no-side --no-candidate 2> $out/bench_fc.err | grep '^{' > $out/rccl_ws1_bench.json
python bench.py --force-collective --overlap off --no-cpu-baseline --no-side --no-candidate 2> /dev/null | grep '^{' > $out/rccl_ws1_no_overlap_bench.json
bash tools/gpu_model.sh xdeepfm > $out/xdeepfm_model.txt 2>&1
bash tools/gpu_model.sh deepfm > $out/deepfm_model.txt 2>&1
python bench.py --workload deepfm --graph > $out/deepfm_config2.json 2> /dev/null
python bench.py --workload dcn --graph --steps 200 --warmup 50 > $out/dcn_c3_bench.json 2> /dev/null
python bench.py --workload fm --graph --steps 200 --warmup 50 > $out/fm_c2_bench.json 2> /dev/null
python bench.py --workload fm --batch 1048576 > $out/fm_bench.json 2> /dev/null
python bench.py --workload autoint --precision f32 --layers 1 > $out/attn_f32_L1_bench.json 2> /dev/null
python tests/cin_error_table.py > $out/cin_error_table.txt 2>&1
head -c 700 gpurun_out/r06_cin/bench.json; echo; cat gpurun_out/r06_cin/mfma_util.txt; cat gpurun_out/r06_cin_bf16x3/bench_profiled.json | head -c 300; echo; cat $out/cin_error_table.txt | tail -8
