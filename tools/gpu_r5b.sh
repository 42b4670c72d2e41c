cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
for v in ${VARIANTS:-pipe nopipe}; do
  export FIL_LIB_PATH=$GRAFT_REPO_ROOT/ml_function_amd/abl/libfil_$v.so
  timeout 300 python bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/r5b/autoint_$v.json 2> gpurun_out/r5b/autoint_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5b/autoint_$v.json"))
print("$v autoint L3 ms/step %.4f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
export FIL_LIB_PATH=$GRAFT_REPO_ROOT/ml_function_amd/abl/libfil_${TESTV:-pipe}.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTK:-(test_attn_at_the_benchmark_shape and f16) or (test_attn_two_waves_per_head and f16)}" > gpurun_out/r5b/test.log 2>&1
tail -5 gpurun_out/r5b/test.log
