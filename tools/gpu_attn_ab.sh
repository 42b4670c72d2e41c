# A/B of experiment builds of the attention kernels: per-kernel times from rocprofv3 (one run per variant)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_ab
export TMPDIR=/tmp
for v in ${VARIANTS}; do
  export FIL_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libfil_$v.so
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python3 bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/attn_ab/autoint_$v.json 2> gpurun_out/attn_ab/autoint_$v.err
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/attn_ab/stats_$v.csv
  python3 - <<PY
import csv,json
d=json.load(open("gpurun_out/attn_ab/autoint_$v.json"))
rows=list(csv.DictReader(open("gpurun_out/attn_ab/stats_$v.csv")))
out=[]
for r in rows:
    n=r["Name"]
    if "attn_" in n:
        short=n.split("(")[0].replace("void fil::","").replace("fil::","")
        out.append("%s %.1fus"%(short, float(r["AverageNs"])/1000))
print("$v ms/step %.4f |"%d["ms_per_step"], " | ".join(out))
PY
done
if [ -n "$TESTV" ]; then
export FIL_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libfil_${TESTV}.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTK:-(test_attn_at_the_benchmark_shape and f16) or (test_attn_two_waves_per_head and f16)}" > gpurun_out/attn_ab/test.log 2>&1
tail -3 gpurun_out/attn_ab/test.log
fi
