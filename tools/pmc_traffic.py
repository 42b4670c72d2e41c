"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of bench.py) into HBM
bytes per launch for every kernel, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
both counters are in KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950, so reads are doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

A kernel instantiation launched p times per step (e.g. the dZ kernel: layer 2 then layer 1) is split into p slots in
dispatch order ("#0" = first launch of a step), given the number of step iterations the profiled run made.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> <iterations = warmup + steps>
(each dir is the -d directory of:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...)
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    """cin_dz3_kernel<1, 20, 128>(...) -> cin_dz3_kernel<1,20,128>"""
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"fil::(\(anonymous namespace\)::)?", "", name)
    return name.replace(" ", "")


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[(short(row["Kernel_Name"]), int(row["Grid_Size"]))].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return {k: sorted(vals) for k, vals in acc.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if not name.startswith(("cin_", "fm_", "dcn_", "attn_", "embed_")):
            continue
        n = max(len(fetch.get(key, [])), len(write.get(key, [])))
        slots = n // iters if iters and n % iters == 0 and n // iters > 1 else 1
        for s in range(slots):
            fk = [v for _, v in fetch.get(key, [])[s::slots]]
            wk = [v for _, v in write.get(key, [])[s::slots]]
            f = sum(fk) / len(fk) if fk else 0.0
            w = sum(wk) / len(wk) if wk else 0.0
            first = (fetch.get(key) or write.get(key))[s][0]
            label = "%s grid=%d" % (name, grid) + (" #%d" % s if slots > 1 else "")
            res[label] = {"first_dispatch": first, "launches": max(len(fk), len(wk)), "fetch_size_kib_raw": f, "write_size_kib": w,
                          "hbm_read_bytes": 2.0 * f * 1024, "hbm_write_bytes": w * 1024, "hbm_bytes": (2.0 * f + w) * 1024}
    with open(out, "w") as fh:
        json.dump({"correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled for gfx950 per MI355X_MICROARCH.md",
                   "per_launch": res}, fh, indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-60s n=%3d read %9.2f MB write %9.2f MB" % (k, v["launches"], v["hbm_read_bytes"] / 1e6, v["hbm_write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
