# A/B of the working tree against a snapshot of HEAD built under _head/ (tools: git archive HEAD ... | tar -x -C _head; build there)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_head
export TMPDIR=/tmp
for v in new head new head; do
  if [ $v = head ]; then cd $GRAFT_REPO_ROOT/_head; else cd $GRAFT_REPO_ROOT; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python3 bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/attn_head/autoint_$v.json 2> $GRAFT_REPO_ROOT/gpurun_out/attn_head/autoint_$v.err
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/attn_head/stats_$v.csv
  python3 - <<PY
import csv,json
d=json.load(open("gpurun_out/attn_head/autoint_$v.json"))
rows=list(csv.DictReader(open("gpurun_out/attn_head/stats_$v.csv")))
out=[]
for r in rows:
    n=r["Name"]
    if "attn_" in n:
        short=n.split("(")[0].replace("void fil::","").replace("fil::","")
        out.append("%s %.1fus"%(short, float(r["AverageNs"])/1000))
print("$v ms/step %.4f |"%d["ms_per_step"], " | ".join(out))
PY
done
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -x -q -m gpu -k "attn or autoint or AutoInt or stack" > gpurun_out/attn_head/test_attn.log 2>&1
tail -5 gpurun_out/attn_head/test_attn.log
