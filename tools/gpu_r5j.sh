cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
for rep in 1 2; do
for t in _r04 .; do
  for w in deepfm xdeepfm; do
    (cd $t && timeout 300 python bench.py --workload $w --graph --steps 50 --warmup 10 2>/dev/null) > gpurun_out/r5j/${w}_$(basename $t)_$rep.json
    python - <<PY
import json
d=json.load(open("gpurun_out/r5j/${w}_$(basename $t)_$rep.json"))
print("$t $w eager %.4f graph %.4f"%(d["ms_per_step"], d["hipgraph_replay_ms_per_step"]))
PY
  done
done
done
