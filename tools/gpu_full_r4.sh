# the driver's round-end sequence: full GPU suite (timed), smoke, default bench
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r4full}
mkdir -p $out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu > $out/test.log 2>&1 ) 2> $out/test.time
grep -v "^$" $out/test.log | tail -5; grep real $out/test.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print("ms/step %.4f value %.0f graph %s roofline %s frac %.3f" % (d["ms_per_step"], d["value"], d.get("hipgraph_replay_ms_per_step"), d["roofline"]["kernel"], d["roofline"]["frac"]))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"))
for k,v in d.get("side_workloads", {}).items(): print(" ", k, v.get("ms_per_step"), v.get("hipgraph_replay_ms_per_step"))
PY
