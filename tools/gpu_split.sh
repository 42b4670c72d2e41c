cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "promotion or split" > gpurun_out/r2s/test_split.log 2>&1
grep -v "^$" gpurun_out/r2s/test_split.log | tail -25
for mode in 2 10; do
  timeout 300 python bench.py --cin-mode $mode --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s/cin_mode$mode.json 2> gpurun_out/r2s/cin_mode$mode.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2s/cin_mode$mode.json"))
print("mode $mode ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
