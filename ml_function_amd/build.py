"""Builds ml_function_amd/libfil_hip.so (gfx950) from csrc/*.hip with hipcc.  In-tree, no JIT cache.

    python -m ml_function_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libfil_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value"] + os.environ.get("FIL_HIPCC_FLAGS", "").split()
# per-source extras: MFMA results in VGPRs for the fused-tail kernels (their 16x16x4 accumulators otherwise get rotated through
# AGPR copies at every h: cin_tail_fwd 0.256 -> 0.249 ms on MI355X)
FILE_FLAGS = {"cin_tail.hip": ["-mllvm", "--amdgpu-mfma-vgpr-form"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "fil.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _flags_changed(obj_dir, flags):
    """The object cache is keyed on source mtimes; the compiler flags (FIL_HIPCC_FLAGS changes codegen) are remembered in a
    stamp file next to the objects, and a mismatch rebuilds everything -- a diagnostic build (e.g. -DFIL_ATTN_STAMPS) can then
    never linger in libfil_hip.so under a later plain build.  Returns (changed, stamp path, wanted contents); build() writes the
    stamp once the rebuild has succeeded."""
    stamp = os.path.join(obj_dir, "flags.stamp")
    want = " ".join(flags)
    try:
        with open(stamp) as fh:
            same = fh.read() == want
    except OSError:
        same = not any(f.endswith(".o") for f in os.listdir(obj_dir))   # fresh directory: nothing stale to distrust
    return (not same), stamp, want


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    changed, stamp, want = _flags_changed(OBJ, FLAGS + [k + "=" + " ".join(v) for k, v in sorted(FILE_FLAGS.items())])
    if changed:
        # objects built with other flags are removed BEFORE anything is recompiled and the stamp is written only after every compile
        # job has succeeded: an interrupted forced rebuild can then never leave old-flag objects behind a matching stamp
        force = True
        for f in os.listdir(OBJ):
            if f.endswith(".o"):
                os.remove(os.path.join(OBJ, f))
        if os.path.exists(stamp):
            os.remove(stamp)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if changed or not os.path.exists(stamp):
        with open(stamp, "w") as fh:
            fh.write(want)
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


ASAN_OBJ = os.path.join(OBJ, "asan")
ASAN_LIB = os.path.join(ASAN_OBJ, "libfil_hip_asan.so")
ASAN_FLAGS = ["-O1", "-g", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-option-ignored",
              "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def asan_runtime():
    """Path of the AddressSanitizer runtime that a non-instrumented python has to LD_PRELOAD to load ASAN_LIB."""
    r = subprocess.run([os.path.join(os.path.dirname(HIPCC), "..", "lib", "llvm", "bin", "clang"), "-print-file-name=libclang_rt.asan-x86_64.so"],
                       capture_output=True, text=True)
    return r.stdout.strip()


def build_asan(verbose=False):
    """The same sources with AddressSanitizer + UBSan on the HOST pass (hipcc ignores the option for the gfx950 device pass:
    GPU sanitizers are not available on this pool) -> build/asan/libfil_hip_asan.so.  tests/test_host.py drives the
    argument-validation / workspace-sizing / error-reporting paths of every entry point through it (no GPU needed)."""
    os.makedirs(ASAN_OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs, objs = [], []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(ASAN_OBJ, src[:-4] + ".o")
        objs.append(o)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + ASAN_FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (asan) failed for %s:\n%s" % (s, r.stderr))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(ASAN_LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", ASAN_LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link (asan) failed:\n%s" % r.stderr)
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
    else:
        print(build(force="--force" in sys.argv))
