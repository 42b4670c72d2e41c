#!/bin/bash
# copies the summaries of tools/profile_r04.sh from gpurun_out/ (scratch) into profiles/ (tracked): bash tools/collect_profiles.sh r04
r=${1:-r04}
for f in bench.json bench_profiled.json kernel_stats.csv mfma_util.json mfma_util.txt pmc_traffic.json pmc_traffic.txt; do
  [ -f gpurun_out/${r}_cin/$f ] && cp gpurun_out/${r}_cin/$f profiles/${r}_cin_$f
done
for f in bench.json kernel_stats.csv pmc_summary.json pmc_summary.txt; do
  [ -f gpurun_out/${r}_attn_f16_L3/$f ] && cp gpurun_out/${r}_attn_f16_L3/$f profiles/${r}_attn_f16_L3_$f
done
for f in bench_profiled.json kernel_stats.csv; do
  [ -f gpurun_out/${r}_cin_bf16x3/$f ] && cp gpurun_out/${r}_cin_bf16x3/$f profiles/${r}_cin_bf16x3_$f
done
[ -f gpurun_out/${r}_cin_bf16x3_sq/mfma_util.json ] && cp gpurun_out/${r}_cin_bf16x3_sq/mfma_util.json profiles/${r}_cin_bf16x3_mfma_util.json
for s in fm_c2 dcn_c3; do
  for f in bench.json kernel_stats.csv pmc_summary.json pmc_summary.txt; do
    [ -f gpurun_out/${r}_$s/$f ] && cp gpurun_out/${r}_$s/$f profiles/${r}_${s}_$f
  done
done
for f in gpurun_out/${r}_misc/*.json gpurun_out/${r}_misc/*.txt; do
  [ -s $f ] && cp $f profiles/${r}_$(basename $f)
done
ls -la profiles/ | grep ${r}_
