cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_suite
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py tests/test_gpu_knobs.py -x -q -m gpu -k "attn or autoint or AutoInt or stack" > gpurun_out/attn_suite/test_attn.log 2>&1
tail -5 gpurun_out/attn_suite/test_attn.log
timeout 300 python bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/attn_suite/autoint_L3.json 2> gpurun_out/attn_suite/autoint_L3.err
python - <<PY
import json
d=json.load(open("gpurun_out/attn_suite/autoint_L3.json"))
print("autoint_L3 ms/step %.4f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
