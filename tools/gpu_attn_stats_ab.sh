# A/B of the backward with and without the forward's saved LayerNorm statistics (FIL_ATTN_SAVE_STATS), one library: per-kernel
# times from rocprofv3, two runs per setting, then the attention tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_stats
export TMPDIR=/tmp
for v in 1 0 1 0; do
  export FIL_ATTN_SAVE_STATS=$v
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python3 bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/attn_stats/autoint_$v.json 2> gpurun_out/attn_stats/autoint_$v.err
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/attn_stats/stats_$v.csv
  python3 - <<PY
import csv,json
d=json.load(open("gpurun_out/attn_stats/autoint_$v.json"))
rows=list(csv.DictReader(open("gpurun_out/attn_stats/stats_$v.csv")))
out=[]
for r in rows:
    n=r["Name"]
    if "attn_" in n:
        short=n.split("(")[0].replace("void fil::","").replace("fil::","")
        out.append("%s %.1fus"%(short, float(r["AverageNs"])/1000))
print("stats=$v ms/step %.4f |"%d["ms_per_step"], " | ".join(out))
PY
done
unset FIL_ATTN_SAVE_STATS
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -x -q -m gpu -k "attn or autoint or AutoInt or stack" > gpurun_out/attn_stats/test_attn.log 2>&1
tail -5 gpurun_out/attn_stats/test_attn.log
