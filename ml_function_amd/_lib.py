"""ctypes binding of libfil_hip.so (include/fil.h).  No CPU fallback: if the library is missing or a call
fails, the product path raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIL_LIB_PATH (development only): an alternative build of the SAME library -- kernel experiments built side by side by
# tools/abl_build.py (-> tools/abl/libfil_<name>.so) and timed against the default one in separate processes.  Honoured only for a
# file inside this repository's tools/abl/ directory: the variable cannot point the product at a library from anywhere else.
# Unset (or anything else) = the in-tree default.
def _lib_path():
    default = os.path.join(_HERE, "libfil_hip.so")
    alt = os.environ.get("FIL_LIB_PATH")
    if not alt:
        return default
    abl = os.path.realpath(os.path.join(_HERE, "..", "tools", "abl")) + os.sep
    if os.path.realpath(alt).startswith(abl):
        return alt
    import warnings
    warnings.warn("FIL_LIB_PATH=%s is outside %s: ignored, loading the in-tree libfil_hip.so" % (alt, abl))
    return default


LIB_PATH = _lib_path()
HEADER_PATH = os.path.join(_HERE, "..", "include", "fil.h")

FIL_F32, FIL_BF16 = 0, 1

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_Z = _c.c_size_t
_F = _c.c_float

# name -> (restype, argtypes); mirrors include/fil.h one to one
SIGNATURES = {
    "fil_version": (_I, []),
    "fil_last_error": (_c.c_char_p, []),
    "fil_profile_begin": (_I, [_c.c_char_p]),
    "fil_profile_end": (_Z, [_c.c_char_p, _Z]),
    "fil_fm_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "fil_fm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fil_fm_pairs_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "fil_fm_pairs_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fil_dcn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fil_dcn_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "fil_dcn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "fil_cin_saved_bytes": (_Z, [_I, _I, _I, _I, _P]),
    "fil_cin_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _P]),
    "fil_cin_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _P]),
    "fil_cin_grad_ready_points": (_I, [_I, _I, _I, _I, _P, _I, _P]),
    "fil_cin_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _Z, _P]),
    "fil_cin_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _Z, _P]),
    "fil_attn_fwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "fil_attn_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "fil_attn_fwd": (_I, [_P] * 10 + [_I, _I, _I, _I, _I, _F, _F, _I, _I, _I, _P, _Z, _P]),
    "fil_attn_bwd": (_I, [_P] * 17 + [_I, _I, _I, _I, _I, _F, _F, _I, _I, _I, _P, _Z, _P]),
    "fil_pattn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "fil_pattn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "fil_embed_gather": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fil_embed_gather_dt": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fil_embed_gather_xt": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fil_embed_scatter_add": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fil_embed_row_ids": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "fil_embed_sort_fields": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _c.c_int64, _P]),
    "fil_embed_segment_sum": (_I, [_P, _P, _P, _P, _P, _P, _c.c_long, _I, _P]),
    "fil_embed_run_sum": (_I, [_P, _P, _P, _P, _c.c_long, _I, _P]),
    "fil_embed_run_sum_dt": (_I, [_P, _P, _P, _P, _c.c_long, _I, _I, _P]),
    "fil_score_add_sigmoid_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "fil_score_add_sigmoid_bwd": (_I, [_P, _P, _P, _I, _P]),
    "fil_bce_mean_fwd": (_I, [_P, _P, _F, _P, _P, _I, _P]),
    "fil_gemm_f32_workspace_bytes": (_Z, [_I, _I, _I]),
    "fil_gemm_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "fil_relu_bias_bwd_workspace_bytes": (_Z, [_I, _I]),
    "fil_relu_bias_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "fil_merge_softmax_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "fil_merge_softmax_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "fil_merge_softmax_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
}


class FilError(RuntimeError):
    pass


_lib = None


def header_symbols():
    """Names of every function declared in include/fil.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fil_[a-z0-9_]+)\s*\(", text)))


def header_abi_version():
    """FIL_ABI_VERSION of include/fil.h (the version this binding's SIGNATURES table was written against)."""
    m = re.search(r"#define\s+FIL_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read())
    if m is None:
        raise FilError("include/fil.h does not define FIL_ABI_VERSION")
    return int(m.group(1))


def load():
    """Loads libfil_hip.so (raises FilError if it has not been built: python -m ml_function_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FilError("libfil_hip.so not found at %s -- build it with `python -m ml_function_amd.build` "
                       "(there is no CPU fallback)" % LIB_PATH)
    # torch first: it ships its own libamdhip64, and the process must end up with ONE HIP runtime -- loaded the other way round
    # (this library pulling in /opt/rocm's copy before torch brings its own) every launch fails with "no ROCm-capable device"
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:
        raise FilError("libfil_hip.so at %s does not export %s -- rebuild it (python -m ml_function_amd.build --force)"
                       % (LIB_PATH, ", ".join(missing)))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    want = header_abi_version()
    if lib.fil_version() != want:
        raise FilError("libfil_hip.so at %s has ABI version %d, include/fil.h says %d -- stale build, rebuild it "
                       "(python -m ml_function_amd.build --force)" % (LIB_PATH, lib.fil_version(), want))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().fil_last_error().decode("utf-8", "replace")
        raise FilError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an integer (the raw getter: no Stream object is built --
    this runs once per C-ABI call)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def profile_begin(filter=None):
    """Start per-kernel event timing; filter = "substr,substr" restricts it to the scopes whose name contains one."""
    load().fil_profile_begin(None if not filter else filter.encode())


def profile_end():
    """Returns {kernel name: dict(count, total_ms, avg_ms, work, executed)} for the launches since profile_begin():
    work = algorithmic flops / bytes per launch (the reference graph's), executed = what the kernels of the scope compute."""
    lib = load()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.fil_profile_end(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, count, ms, work, executed = line.split()
        out[name] = dict(count=int(count), total_ms=float(ms), avg_ms=float(ms) / max(int(count), 1), work=float(work),
                         executed=float(executed))
    return out
