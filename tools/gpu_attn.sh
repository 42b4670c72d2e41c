cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu -k "attn or autoint" > gpurun_out/r2a/test_attn.log 2>&1
grep -v "^$" gpurun_out/r2a/test_attn.log | tail -12
for prec in f16_mfma f32; do
  for L in 1 3; do
    timeout 300 python bench.py --workload autoint --precision $prec --layers $L --steps 10 --warmup 3 > gpurun_out/r2a/autoint_${prec}_L$L.json 2> gpurun_out/r2a/autoint_${prec}_L$L.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2a/autoint_${prec}_L$L.json"))
print("$prec L=$L ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
  done
done
