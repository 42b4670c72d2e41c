"""Lists the suspicious `s_waitcnt vmcnt(N)` of the MFMA loops of a HIP source file ("keeping the operand
queues in flight"): compiles the file to gfx950 assembly and, for every loop that holds at least --min-mfma MFMAs, prints the
wait counts at or below --max-count together with the loop's size, MFMA and load counts.  A streaming kernel that keeps a
DEPTH-deep operand queue should only show counts near DEPTH; vmcnt(0..3) inside such a loop means the queue is drained there.

    python tools/waitcnt_scan.py ml_function_amd/csrc/cin_fwd.hip --kernel fwd3_kernelILi2ELi20ELb0ELb0
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile


def compile_to_asm(src, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S", src,
           "-o", out] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    return out


def scan(asm_path, pats, min_mfma, max_count):
    lines = open(asm_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[A-Za-z0-9_]*:", l)]
    for idx, (i, name) in enumerate(starts):
        if pats and not any(p in name for p in pats):
            continue
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        body = lines[i:end]
        labels = {}
        for j, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = j
        seen = set()
        for j, l in enumerate(body):
            m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < j):
                continue
            a = labels[m.group(1)]
            if a in seen:
                continue
            seen.add(a)
            seg = body[a:j + 1]
            n_mfma = sum("v_mfma" in x for x in seg)
            if n_mfma < min_mfma:
                continue
            n_load = sum(("global_load" in x or "buffer_load" in x) for x in seg)
            low = []
            for t, x in enumerate(seg):
                w = re.search(r"vmcnt\((\d+)\)", x)
                if w and int(w.group(1)) <= max_count:
                    low.append("+%d:vmcnt(%s)" % (t, w.group(1)))
            print("%-70s loop@%-5d lines %-5d mfma %-4d loads %-4d %s" % (name[:70], a, len(seg), n_mfma, n_load,
                                                                          " ".join(low) if low else "-"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("source", help=".hip file (or an already generated .s)")
    ap.add_argument("--kernel", action="append", default=[], help="substring of the mangled kernel name (repeatable)")
    ap.add_argument("--min-mfma", type=int, default=16)
    ap.add_argument("--max-count", type=int, default=3)
    ap.add_argument("--flag", action="append", default=[], help="extra hipcc flag (repeatable)")
    a = ap.parse_args()
    asm = a.source if a.source.endswith(".s") else compile_to_asm(os.path.abspath(a.source), a.flag)
    scan(asm, a.kernel, a.min_mfma, a.max_count)


if __name__ == "__main__":
    main()
