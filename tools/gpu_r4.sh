# round-4 iteration script: quick CIN parity subset + bench (usage: gpurun -- 'bash tools/gpu_r4.sh <tag> [pytest -k expr]')
cd $GRAFT_REPO_ROOT
tag=${1:-r4}
kexpr=${2:-"test_cin and not promotion and not benchmark_shape and not large_batch and not split"}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$kexpr" > $out/test.log 2>&1
grep -v "^$" $out/test.log | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print("ms/step %.4f  value %.0f" % (d["ms_per_step"], d["value"]))
for k,v in sorted(d["kernels"].items()): print("  %-18s %.4f" % (k, v["avg_ms"]))
PY
