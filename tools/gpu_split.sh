cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "promotion or split or bench_shape or cin_modes" > gpurun_out/r2s/test_split.log 2>&1
grep -v "^$" gpurun_out/r2s/test_split.log | tail -25
for dzmb in 1 2; do
  FIL_CIN_DZ_MB=$dzmb timeout 300 python bench.py --cin-mode 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2s/cin_mode2_dz$dzmb.json 2> gpurun_out/r2s/cin_mode2_dz$dzmb.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r2s/cin_mode2_dz$dzmb.json"))
print("mode 2 DZ_MB=$dzmb ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
