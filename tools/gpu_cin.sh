cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_properties.py -x -q -m gpu -k "cin or CIN or xdeepfm" > gpurun_out/r2c/test_cin.log 2>&1
grep -v "^$" gpurun_out/r2c/test_cin.log | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2c/bench.json"))
print("mode 0 ms/step %.3f"%d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2c
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.log
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json 6 > $out/pmc_traffic.txt
grep -E "dz3|fwd3|dw3" $out/pmc_traffic.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
