cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fm
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_models_gpu.py -x -q -m gpu -k "fm or FM or deepfm" 2>&1 | tail -4
for B in 4096 1048576; do
  timeout 300 python bench.py --workload fm --batch $B --steps 20 --warmup 5 > gpurun_out/fm/fm_$B.json 2> gpurun_out/fm/fm_$B.err
  python - <<PY
import json
d=json.load(open("gpurun_out/fm/fm_$B.json"))
print("FM B=$B ms/step %.4f"%d["ms_per_step"], d["roofline"]["kernel"], "frac %.3f"%d["roofline"]["frac"], {k:(v["avg_ms"], round(v["work"]/v["avg_ms"]/1e6,1)) for k,v in d["kernels"].items()})
PY
done
