cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
for f in 1 0; do
  FIL_CIN_FORK=$f timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5g/cin_fork$f.json 2> gpurun_out/r5g/cin_fork$f.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5g/cin_fork$f.json"))
print("fork=$f cin ms/step %.4f graph %s gpu_kernel %.4f"%(d["ms_per_step"], d.get("hipgraph_replay_ms_per_step"), d["gpu_kernel_ms_per_step"]), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_gpu.py -x -q -m gpu -k "test_cin_at_the_benchmark_shape or test_dp or capturable or graph or test_cin_fused_tail or embed" > gpurun_out/r5g/test.log 2>&1
tail -4 gpurun_out/r5g/test.log
