cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
bash tools/gpu_attn.sh 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cin" > gpurun_out/r2b/test_cin.log 2>&1; tail -3 gpurun_out/r2b/test_cin.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; tail -2 gpurun_out/r2b/bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --force-collective --no-cpu-baseline > gpurun_out/r2b/bench_fc.json 2> gpurun_out/r2b/bench_fc.err; tail -2 gpurun_out/r2b/bench_fc.err
timeout 600 python bench.py --steps 10 --warmup 3 --force-collective --no-overlap --no-cpu-baseline > gpurun_out/r2b/bench_fc_noov.json 2> gpurun_out/r2b/bench_fc_noov.err; tail -2 gpurun_out/r2b/bench_fc_noov.err
python - <<PY
import json
for n in ["bench","bench_fc","bench_fc_noov"]:
    try:
        d=json.load(open("gpurun_out/r2b/%s.json"%n)); print(n, "value %.0f ms %.3f"%(d["value"], d["ms_per_step"]), d.get("rccl"))
    except Exception as e: print(n, "ERR", e)
PY
