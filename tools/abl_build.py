"""Side-by-side experiment builds: python tools/abl_build.py <name> "<extra hipcc flags>" <file.hip> [...]
Recompiles only the listed sources of ml_function_amd/csrc with the extra flags and links them with the default build's other
objects into tools/abl/libfil_<name>.so (git-ignored, outside the product package; travels with gpurun).  Run with FIL_LIB_PATH=<that file>."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ml_function_amd import build as B


def main():
    name, flags, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    B.build(verbose=False)
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abl")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B._sources():
        o = os.path.join(B.OBJ, src[:-4] + ".o")
        if src in files:
            o = os.path.join(out_dir, "%s_%s.o" % (name, src[:-4]))
            cmd = [B.HIPCC] + B.FLAGS + B.FILE_FLAGS.get(src, []) + flags + ["-c", os.path.join(B.CSRC, src), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stderr)
        objs.append(o)
    lib = os.path.join(out_dir, "libfil_%s.so" % name)
    r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    print(lib)


if __name__ == "__main__":
    main()
