cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/grp
for G in 1 2 4; do
  rm -f ml_function_amd/build/attn.o
  FIL_HIPCC_FLAGS="-DFIL_ATTN_TILE_GROUP=$G" python -m ml_function_amd.build > gpurun_out/grp/build_$G.log 2>&1
  for L in 1 3; do
    timeout 300 python bench.py --workload autoint --precision f16_mfma --layers $L --steps 10 --warmup 3 > gpurun_out/grp/g${G}_L$L.json 2> /dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/grp/g${G}_L$L.json"))
print("GROUP=$G L=$L ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
  done
done
