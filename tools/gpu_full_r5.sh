# the whole GPU suite as the driver runs it + smoke + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full_r5
( time timeout 1700 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/full_r5/pytest_gpu.log 2>&1
tail -6 gpurun_out/full_r5/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_r5/smoke.log 2>&1; tail -2 gpurun_out/full_r5/smoke.log
python bench.py > gpurun_out/full_r5/bench.json 2> gpurun_out/full_r5/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/full_r5/bench.json"))
print("cin ms/step %.4f windows %s graph %.4f"%(d["ms_per_step"], [round(w,4) for w in d["ms_per_step_windows"]], d["hipgraph_replay_ms_per_step"]))
print({k:(round(v.get("ms_per_step"),4),v.get("hipgraph_replay_ms_per_step")) for k,v in d["side_workloads"].items()})
print(d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
