"""MFMA-pipe utilisation per kernel launch from one rocprofv3 SQ-counter pass of bench.py.

usage: python tools/mfma_util.py <sq_dir> <out.json> <iterations>
<sq_dir> is the -d directory of
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
              --kernel-trace --output-format csv -d <sq_dir> -- python3 bench.py ...      (its own run: counters + kernel-trace only)
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles summed over the 1024 SIMDs over the
kernel's own cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back").  Launch slots of one
instantiation (layer 2 then layer 1 in the backward) are separated by dispatch order like tools/pmc_traffic.py does, and the
CIN GEMM kernels are also reported under bench.py's profiler scope names (`by_scope`), which is what bench.py looks up.
"""
import json
import sys

from pmc_traffic import per_kernel

COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES",
            "GRBM_GUI_ACTIVE"]


def scopes_of(res):
    """bench.py's profiler scope -> the counters of the GEMM launch behind it (bench._gemm_launch_of: forward l1, .., tail; backward
    tail, .., l1; '#slot' entries for a kernel launched several times per step, as the quadratic tail does)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    def pick(k):
        return {"kernel": k, "mfma_busy_frac": res[k].get("mfma_busy_frac"),
                "wave_cycles_waiting_frac": res[k].get("wave_cycles_waiting_frac"),
                "wave_cycles_issue_stalled_frac": res[k].get("wave_cycles_issue_stalled_frac")}
    by_scope = {}
    base = lambda k: k.split(" #")[0]
    for kind in ("fwd", "bwd_dz", "bwd_dw"):
        first, last = bench._gemm_launch_of(res, kind, "l1"), bench._gemm_launch_of(res, kind, "tail")
        # a tail is in use when the last GEMM of the chain is a cin_tail_* kernel (fused tail) or a second launch of the first
        # layer's kernel (quadratic tail); otherwise the chain is l1, l2, ..
        has_tail = last is not None and (base(last).startswith("cin_tail_") or (last != first and base(last) == base(first)))
        n = 1
        while True:
            k = bench._gemm_launch_of(res, kind, "l%d" % n)
            if k is None or (has_tail and k == last):
                break
            by_scope["cin_%s_l%d" % (kind, n)] = pick(k)
            n += 1
        if has_tail:
            by_scope["cin_%s_tail" % kind] = pick(last)
        q = bench._gemm_launch_of(res, kind, "q")   # merged quadratic tail: cin_fwdq_kernel / cin_dwq_kernel / cin_dz2_kernel
        if q is not None:
            by_scope["cin_%s_q" % kind] = pick(q)
    return by_scope

def main():
    d, out, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
    data = {c: per_kernel(d, c) for c in COUNTERS}
    res = {}
    for key in sorted(data["GRBM_GUI_ACTIVE"]):
        name, grid = key
        if not name.startswith(("cin_", "fm_", "dcn_", "attn_", "embed_")):
            continue
        n = len(data["GRBM_GUI_ACTIVE"][key])
        slots = n // iters if iters and n % iters == 0 and n // iters > 1 else 1
        for s in range(slots):
            e = {"first_dispatch": data["GRBM_GUI_ACTIVE"][key][s][0], "launches": len(data["GRBM_GUI_ACTIVE"][key][s::slots])}
            for c in COUNTERS:
                v = [x for _, x in data[c].get(key, [])[s::slots]]
                e[c] = sum(v) / len(v) if v else None
            gui = e["GRBM_GUI_ACTIVE"]
            if gui and e["SQ_VALU_MFMA_BUSY_CYCLES"] is not None:
                e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0)
            wc = e["SQ_WAVE_CYCLES"]
            if wc:
                e["wave_cycles_waiting_frac"] = (e["SQ_WAIT_ANY"] or 0.0) / wc
                e["wave_cycles_issue_stalled_frac"] = (e["SQ_WAIT_INST_ANY"] or 0.0) / wc
                e["wave_cycles_issuing_frac"] = (e["SQ_ACTIVE_INST_ANY"] or 0.0) / wc
            res["%s grid=%d" % (name, grid) + (" #%d" % s if slots > 1 else "")] = e
    by_scope = scopes_of(res)
    with open(out, "w") as fh:
        json.dump({"definition": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "by_scope": by_scope,
                   "per_launch": res}, fh, indent=1, sort_keys=True)
    for k, v in by_scope.items():
        print("%-16s %-50s mfma busy %.3f  waiting %.3f  issue-stalled %.3f" % (k, v["kernel"], v["mfma_busy_frac"] or 0,
                                                                                v["wave_cycles_waiting_frac"] or 0, v["wave_cycles_issue_stalled_frac"] or 0))


if __name__ == "__main__":
    main()
