# ablation builds on the GPU box: usage gpurun -- 'bash tools/gpu_abl.sh tag "-DFLAG1 -DFLAG2" ...' (pairs); bench only
cd $GRAFT_REPO_ROOT
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p gpurun_out/$tag
  FIL_HIPCC_FLAGS="$flags" python -m ml_function_amd.build > gpurun_out/$tag/build.log 2>&1 || { tail -5 gpurun_out/$tag/build.log; continue; }
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench.json"))
print("$tag [$flags] ms/step %.4f" % d["ms_per_step"], " ".join("%s=%.4f" % (k.replace("cin_",""), v["avg_ms"]) for k,v in sorted(d["kernels"].items()) if v["avg_ms"] > 0.05))
PY
done
