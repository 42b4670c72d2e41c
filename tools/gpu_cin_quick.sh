cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_properties.py -x -q -m gpu -k "cin or CIN or xdeepfm" > gpurun_out/r2c/test_cin.log 2>&1
grep -v "^$" gpurun_out/r2c/test_cin.log | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2c/bench.json"))
print("mode 0 ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
c=d.get("candidate_split_bf16")
if c: print("split ms/step %.3f"%c["ms_per_step"], c["kernels_ms"])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2c/stats -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2> gpurun_out/r2c/stats.log
find gpurun_out/r2c/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2c/kernel_stats.csv
find gpurun_out/r2c -name "*kernel_trace.csv" -delete
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r2c/kernel_stats.csv')))
for r in rows[:34]:
    if float(r['AverageNs'])<80000: print("%-70s calls %5s avg %7.1f us"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
