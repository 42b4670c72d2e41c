#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcta
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --cin-mode 0 > /dev/null 2> $out/p1.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void fil::","").split("(")[0]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    if "cin_" not in n or "true>" in n.replace(" ", "") and "dzs" not in n: pass
    gui = sum(c["GRBM_GUI_ACTIVE"]) / max(len(c["GRBM_GUI_ACTIVE"]), 1) / 8.0
    if gui < 20000: continue
    row = {k: sum(v) / len(v) for k, v in c.items()}
    print("%-60s cyc %8.0f  TA_BUSY_avr/cyc %.3f  TA_stall_by_TC/cyc/256 %.3f  TCP_pending/cyc/256 %.3f  tagconf/cyc/256 %.3f  rd_waves %d" % (
        n[:60], gui, row.get("TA_BUSY_avr", 0) / gui, row.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / gui / 256,
        row.get("TCP_PENDING_STALL_CYCLES_sum", 0) / gui / 256, row.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / gui / 256, row.get("TA_FLAT_READ_WAVEFRONTS_sum", 0)))
PY
tail -3 $out/p1.log
