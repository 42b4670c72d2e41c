cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_suite
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -x -q -m gpu -k "attn or autoint or AutoInt or stack" > gpurun_out/attn_suite/test_attn.log 2>&1
tail -5 gpurun_out/attn_suite/test_attn.log
timeout 300 python bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/attn_suite/autoint_L3.json 2> gpurun_out/attn_suite/autoint_L3.err
timeout 300 python bench.py --workload autoint --precision f32 --layers 1 --steps 10 --warmup 3 > gpurun_out/attn_suite/autoint_f32_L1.json 2> gpurun_out/attn_suite/autoint_f32_L1.err
python - <<PY
import json
for n in ("autoint_L3","autoint_f32_L1"):
    d=json.load(open("gpurun_out/attn_suite/%s.json"%n))
    print(n,"ms/step %.4f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
