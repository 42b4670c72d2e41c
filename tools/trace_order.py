#!/usr/bin/env python
"""Launch order of ONE benchmark step from a rocprofv3 --kernel-trace CSV: every kernel between two consecutive
cin_head_bwd launches, in start-time order, with its queue, start (us after the first) and duration -- shows the RCCL
all-reduce kernels of bench.py --force-collective running on their own queue underneath the backward's kernels.

    python tools/trace_order.py <dir with *kernel_trace.csv> [occurrence of cin_head_bwd to start at, default 3]
"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under " + d)
    rows = []
    with open(files[0]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "cin_head_bwd" in r[2]]
    if len(marks) < skip + 2:
        sys.exit("fewer than %d steps in the trace" % (skip + 2))
    # from the forward in front of this backward: back up to the previous transpose_out / start
    lo, hi = marks[skip], marks[skip + 1]
    t0 = rows[lo][0]
    print("# one step: kernels from one cin_head_bwd launch to the next (start us, duration us, queue, stream, name)")
    for s, e, name, q, st in rows[lo:hi]:
        short = name.split("(")[0].replace("void ", "").replace("fil::", "")
        tag = "  <-- RCCL" if ("ccl" in name.lower() or "allreduce" in name.lower()) else ""
        print("%10.1f %9.1f  q%-3s s%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, short[:110], tag))


if __name__ == "__main__":
    main()
