# A/B of experiment builds (tools/abl_build.py): usage gpurun -- 'bash tools/gpu_libs.sh tag name1 name2 ...'  ("default" = the in-tree library)
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib=$GRAFT_REPO_ROOT/tools/abl/libfil_$name.so
  FIL_LIB_PATH=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side $BENCH_EXTRA > gpurun_out/$tag/bench_$name.json 2> gpurun_out/$tag/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$tag/bench_$name.json"))
    print("%-10s ms/step %.4f " % ("$name", d["ms_per_step"]), " ".join("%s=%.4f" % (k.replace("cin_",""), v["avg_ms"]) for k,v in sorted(d["kernels"].items()) if v["avg_ms"] > float("${THRESH:-0.04}")))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/$tag/bench_$name.err").read()[-800:])
PY
done
