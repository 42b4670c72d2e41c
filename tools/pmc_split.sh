#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the split-bf16 step (bench.py --cin-mode 2), summarised by tools/mfma_util.py
set -u
tag=${1:-r06_split_sq}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --cin-mode 2 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-side --no-graph-replay > /dev/null 2> $out/pmc_sq.log
python3 tools/mfma_util.py $out/pmc_sq $out/mfma_util.json 12 > $out/mfma_util.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq2 -- python3 bench.py --cin-mode 2 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-side --no-graph-replay > /dev/null 2> $out/pmc_sq2.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/pmc_sq2/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "_b_kernel" in n:
            acc[n.split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in v.items()})
PY
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
grep -E "_b_kernel" -A0 $out/mfma_util.txt | head -20
cat $out/mfma_util.txt | head -40
