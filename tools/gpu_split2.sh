cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "roctx" 2>&1 | tail -3
for mb in 1 2; do
  FIL_CIN_MB=$mb timeout 300 python bench.py --cin-mode 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s/cin_mode2_mb$mb.json 2> gpurun_out/r2s/cin_mode2_mb$mb.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2s/cin_mode2_mb$mb.json"))
print("mode 2 FIL_CIN_MB=$mb ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if 'l1' in k or 'l2' in k})
PY
done
