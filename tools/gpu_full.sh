#!/bin/bash
# the whole -m gpu suite + the default bench line (what the driver runs at round end) + the per-kernel table
cd $GRAFT_REPO_ROOT
tag=${1:-full}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q -s > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -E "c5 at B|passed|failed|^FAILED|rc=" $out/pytest.log | cut -c1-400
python bench.py > $out/bench.json 2> $out/bench.err
python - <<PY
import json
j=json.loads([l for l in open("$out/bench.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], "executed_frac", j.get("executed_frac"))
print("roofline", {k: j["roofline"][k] for k in ("kernel","achieved","frac","avg_launch_ms")})
for k,v in j["kernels"].items(): print("  ", k, v.get("avg_ms"), v.get("executed_tflops"))
print("cpu", j.get("cpu_baseline"))
print("side", json.dumps(j.get("side_workloads"), indent=1)[:3000])
PY
