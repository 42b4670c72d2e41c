cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
./tools/bin/probe_tile > gpurun_out/r5a/probe_tile.txt 2>&1
cat gpurun_out/r5a/probe_tile.txt
timeout 300 python bench.py --workload autoint --precision f16_mfma --layers 3 --steps 20 --warmup 5 > gpurun_out/r5a/autoint_L3.json 2> gpurun_out/r5a/autoint_L3.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5a/autoint_L3.json"))
print("autoint L3 ms/step %.4f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/cin.json 2> gpurun_out/r5a/cin.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5a/cin.json"))
print("cin ms/step %.4f"%d["ms_per_step"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()})
print({k:(v.get("ms_per_step"),v.get("hipgraph_replay_ms_per_step")) for k,v in d["side_workloads"].items()})
PY
