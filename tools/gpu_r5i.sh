cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-side > gpurun_out/r5i/bench$i.json 2> gpurun_out/r5i/bench$i.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5i/bench$i.json"))
print("cin ms/step %.4f windows %s graph %.4f dz %.4f"%(d["ms_per_step"], [round(w,4) for w in d["ms_per_step_windows"]], d["hipgraph_replay_ms_per_step"], d["roofline"]["avg_launch_ms"]))
PY
done
