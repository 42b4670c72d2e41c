"""Per-kernel averages of rocprofv3 PMC passes.

usage: python tools/pmc_summary.py <out.json> <dir> [<dir> ...]
Each <dir> is the -d directory of one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 ...`
pass (counters are collected in SEPARATE passes; never combined with --stats or trace domains, see tools/profile_attn.sh).
Output: {kernel: {"dispatches": n, counter: mean value per dispatch, ...}} plus derived fractions when their inputs exist:
  valu_busy  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES... (all SQ_* cycle counters are summed over waves, in quad-cycles)
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"fil::(\(anonymous namespace\)::)?", "", name)
    return name.replace(" ", "")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in sorted(acc.items()):
        if not (k.startswith("attn_") or k.startswith("cin_") or k.startswith("fm_") or k.startswith("dcn_") or k.startswith("embed_")):
            continue
        e = {"dispatches": max(len(v) for v in cs.values())}
        for cn, v in sorted(cs.items()):
            e[cn] = sum(v) / len(v)
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for cn, label in [("SQ_WAIT_ANY", "frac_wave_cycles_waiting"), ("SQ_WAIT_INST_ANY", "frac_wave_cycles_issue_stalled"),
                              ("SQ_ACTIVE_INST_ANY", "frac_wave_cycles_issuing"), ("SQ_ACTIVE_INST_VALU", "frac_wave_cycles_valu"),
                              ("SQ_ACTIVE_INST_LDS", "frac_wave_cycles_lds"), ("SQ_WAIT_INST_LDS", "frac_wave_cycles_lds_stalled")]:
                if cn in e:
                    e[label] = e[cn] / wc
        if "SQ_BUSY_CYCLES" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
            # SQ_BUSY_CYCLES: per-SE busy cycles summed (quad-cycles x ...); reported raw, the ratio is only indicative
            e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            # MI355X_MICROARCH.md (HBM): both in KiB; FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> doubled
            e["hbm_bytes"] = (2 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024
        res[k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, e in res.items():
        print(k)
        for cn, v in e.items():
            print("   %-36s %s" % (cn, ("%.4g" % v) if isinstance(v, float) else v))


if __name__ == "__main__":
    main()
