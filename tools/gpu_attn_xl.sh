cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/xl
for X in 2 4; do
  rm -f ml_function_amd/build/attn.o
  FIL_HIPCC_FLAGS="-DFIL_ATTN_XL_MAXNC=$X" python -m ml_function_amd.build > gpurun_out/xl/build_$X.log 2>&1
  timeout 300 python bench.py --workload autoint --precision f16_mfma --layers 3 --steps 10 --warmup 3 > gpurun_out/xl/x${X}_L3.json 2> gpurun_out/xl/x${X}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/xl/x${X}_L3.json"))
print("XLMAX=$X L=3 ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
