# experiment: waves-per-EU budget of the fused attention backward (register cap -> compiler spills cold values)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wpe
for W in 2 3 4; do
  rm -f ml_function_amd/build/attn.o
  FIL_HIPCC_FLAGS="-DFIL_ATTN_BWD_WPE(NC,F16)=$W" python -m ml_function_amd.build > gpurun_out/wpe/build_$W.log 2>&1
  for L in 1 3; do
    timeout 300 python bench.py --workload autoint --precision f16_mfma --layers $L --steps 10 --warmup 3 > gpurun_out/wpe/w${W}_L$L.json 2> gpurun_out/wpe/w${W}_L$L.err
    python - <<PY
import json
d=json.load(open("gpurun_out/wpe/w${W}_L$L.json"))
print("WPE=$W L=$L ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attn" 2>&1 | tail -3
