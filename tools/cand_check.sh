# candidate (split-bf16) timing beside the headline vs as the main workload, on one box
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-side 2>/dev/null > gpurun_out/cand_default.json
python bench.py --no-cpu-baseline --no-side --cin-mode 2 --windows 3 2>/dev/null > gpurun_out/cand_mode2.json
python - <<PY
import json
d=json.load(open("gpurun_out/cand_default.json")); print("default: exact", [round(x,4) for x in d["ms_per_step_windows"]], "replay", d.get("hipgraph_replay_ms_per_step"), "cand", d["candidate_bf16x3"]["ms_per_step"], d["candidate_bf16x3"].get("hipgraph_replay_ms_per_step"))
d=json.load(open("gpurun_out/cand_mode2.json")); print("mode2 main:", [round(x,4) for x in d["ms_per_step_windows"]], "replay", d.get("hipgraph_replay_ms_per_step"))
PY
