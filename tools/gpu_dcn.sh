cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dcn
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -x -q -m gpu -k "dcn or DCN" 2>&1 | tail -4
for B in 8192 131072; do
  timeout 300 python bench.py --workload dcn --batch $B --steps 20 --warmup 5 --graph > gpurun_out/dcn/dcn_$B.json 2> gpurun_out/dcn/dcn_$B.err
  python - <<PY
import json
d=json.load(open("gpurun_out/dcn/dcn_$B.json"))
print("DCN B=$B ms/step %.4f graph %.4f"%(d["ms_per_step"], d["hipgraph_replay_ms_per_step"] or 0), d["roofline"]["kernel"], "frac %.3f"%d["roofline"]["frac"], {k:(v["avg_ms"], round(v["work"]/v["avg_ms"]/1e6,1)) for k,v in d["kernels"].items()})
PY
done
