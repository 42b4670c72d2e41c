# the fused score head / loss: its parity tests, the whole-model tests, and the two whole-model bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/head
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_models_gpu.py -x -q -m gpu -k "score or crossentropy or models_gpu or xdeepfm or XDeepFM or DeepFM or train or oracle_composition or adam" > gpurun_out/head/test.log 2>&1
tail -5 gpurun_out/head/test.log
for w in xdeepfm deepfm; do
  python bench.py --workload $w --graph > gpurun_out/head/$w.json 2> gpurun_out/head/$w.err
  python - <<PY
import json
d=json.load(open("gpurun_out/head/$w.json"))
print("$w eager %.4f replay %.4f"%(d["ms_per_step"], d.get("hipgraph_replay_ms_per_step") or -1))
PY
done
bash tools/gpu_model.sh xdeepfm > gpurun_out/head/xdeepfm_model.txt 2>&1; head -3 gpurun_out/head/xdeepfm_model.txt | cut -c1-300
