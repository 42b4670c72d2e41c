cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fm
for cfg in "8192 2" "16384 2" "8192 1" "16384 1" "32768 1" "4096 1"; do
  set -- $cfg
  FIL_FM_TILE_BYTES=$1 FIL_FM_NBUF=$2 timeout 300 python bench.py --workload fm --batch 1048576 --steps 20 --warmup 5 > gpurun_out/fm/sw_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/fm/sw_$1_$2.json"))
print("tile $1 nbuf $2:", {k:(v["avg_ms"], round(v["work"]/v["avg_ms"]/1e6,1)) for k,v in d["kernels"].items()})
PY
done
