cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knobs.py tests/test_models_gpu.py tests/test_data_prepare.py -x -q -m gpu -k "embed or knob or cross_layer_outside or test_cin_fused_tail or model or xdeepfm or deepfm or train or adam or data_prepare" > gpurun_out/r5h/test.log 2>&1
tail -4 gpurun_out/r5h/test.log
for w in xdeepfm deepfm; do
timeout 300 python bench.py --workload $w --graph --steps 50 --warmup 10 > gpurun_out/r5h/$w.json 2> gpurun_out/r5h/$w.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5h/$w.json"))
print("$w eager %.4f graph %.4f"%(d["ms_per_step"], d["hipgraph_replay_ms_per_step"]))
PY
done
