"""Every C-ABI entry point driven through its argument-validation, shape-limit, workspace-sizing and error-reporting paths
WITHOUT a GPU (all of them return before the first launch).  Run in-process against libfil_hip.so by tests/test_host.py
and, as a script, against the AddressSanitizer + UBSan build of the same sources:

    LD_PRELOAD=<libclang_rt.asan> python tests/host_calls.py ml_function_amd/build/asan/libfil_hip_asan.so
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_function_amd import _lib  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def run(lib):
    n = 0

    def expect(rc, want, needle=None):
        nonlocal n
        n += 1
        assert rc == want, (n, rc, want, lib.fil_last_error())
        if needle is not None:
            assert needle in lib.fil_last_error(), (n, lib.fil_last_error())

    assert lib.fil_version() == _lib.header_abi_version()
    H3 = _lib.int_array([128, 128, 128])
    # FM / DCN
    expect(lib.fil_fm_fwd(None, None, None, 4, 3, 2, 0, None), -1, b"bad argument")
    expect(lib.fil_fm_fwd(None, None, None, 0, 3, 2, 0, None), 0)
    expect(lib.fil_fm_fwd(None, None, None, 4, 3, 2, 7, None), -1)
    expect(lib.fil_fm_bwd(None, None, None, None, 4, 3, 2, 0, None), -1)
    expect(lib.fil_fm_pairs_fwd(None, None, 4, 1, 2, None), -1)
    expect(lib.fil_fm_pairs_bwd(None, None, None, 0, 3, 2, None), 0)
    expect(lib.fil_dcn_fwd(None, None, None, None, None, 4, 5000, 3, None), -1)        # D > 4096: the generic kernels take it (argument check next)
    expect(lib.fil_dcn_fwd(None, None, None, None, None, 4, 100, 7, None), -1)
    expect(lib.fil_dcn_fwd(None, None, None, None, None, 4, 100, 17, None), -4, b"16")
    expect(lib.fil_dcn_bwd(None, None, None, None, None, None, None, None, 4, 100, 17, None, 0, None), -4, b"16")
    assert lib.fil_dcn_bwd_workspace_bytes(8, 6400, 8) > 0
    expect(lib.fil_dcn_bwd(None, None, None, None, None, None, None, None, 4, 100, 3, None, 0, None), -1)
    assert lib.fil_dcn_bwd_workspace_bytes(8192, 1248, 3) > 0 and lib.fil_dcn_bwd_workspace_bytes(0, 1248, 3) == 0
    assert lib.fil_dcn_bwd_workspace_bytes(16, 3200, 2) > 0          # generic two-pass path
    # CIN
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 9, H3, 1, 0, None, 0, None), -4)
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 1, _lib.int_array([300]), 1, 0, None, 0, None), -4)
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 3, H3, 1, 1999, None, 0, None), -4, b"mode")
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 3, H3, 1, 1024, None, 0, None), -4, b"mode 1024")   # beyond the mode bits
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 3, H3, 1, 2, None, 0, None), -1, b"")   # FIL_CIN_BF16X3 is a mode again (ABI 214): argument check next
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 70, 16, 3, H3, 1, 0, None, 0, None), -4, b"64 fields")
    expect(lib.fil_cin_bwd(None, None, None, None, None, None, None, None, None, None, None, None, 4, 39, 16, 3, H3, 1, 0, None, None, 0, None), -1)
    for B in (0, 1, 4096, 150000):
        assert lib.fil_cin_saved_bytes(B, 39, 16, 3, H3) >= 0
        assert lib.fil_cin_fwd_workspace_bytes(B, 39, 16, 3, H3) >= 0
        assert lib.fil_cin_bwd_workspace_bytes(B, 39, 16, 3, H3) >= 0
    # xT | first layer's map | quadratic tail: R [M][128], T [F*F][128], wsum_L [128*F], cvec [128] (rounded up to 256 bytes)
    # (+ wsum_p [128*F], its MFMA operand copy [40*128], T in the dZ kernel's slot order [26 tiles * 32 * 128])
    qt = (65536 * 128 + 39 * 39 * 128 + 128 * 39 + 128 + 128 * 39 + 40 * 128 + 26 * 32 * 128) * 4
    assert lib.fil_cin_saved_bytes(4096, 39, 16, 3, H3) == 65536 * 39 * 4 + 65536 * 128 * 4 + (qt + 255) // 256 * 256
    pts = (ctypes.c_int * 4)()
    assert lib.fil_cin_grad_ready_points(4096, 39, 16, 3, H3, 0, pts) == 2 and list(pts) == [0, 1, 1, 0]     # merged quadratic tail: layer 0 (+ the head), then 1 + 2
    assert lib.fil_cin_grad_ready_points(4096, 39, 16, 3, H3, 512, pts) == 3 and list(pts) == [2, 1, 1, 0]   # FIL_CIN_NOQMERGE: layers 1, 2 together first
    assert lib.fil_cin_grad_ready_points(4096, 39, 16, 3, H3, 32, pts) == 4 and list(pts) == [3, 2, 1, 0]    # FIL_CIN_NOTAIL
    assert lib.fil_cin_grad_ready_points(0, 39, 16, 3, H3, 0, pts) == 1 and list(pts) == [0, 0, 0, 0]
    assert lib.fil_cin_grad_ready_points(4096, 39, 16, 3, H3, 0, None) == -1
    assert lib.fil_cin_grad_ready_points(4096, 65, 16, 3, H3, 0, pts) == -4
    # attention
    nul9 = [None] * 10
    expect(lib.fil_attn_fwd(*nul9, 4, 200, 16, 4, 32, 0.25, 1e-3, 1, 0, 0, None, 0, None), -4)
    expect(lib.fil_attn_fwd(*nul9, 4, 200, 16, 4, 16, 0.25, 1e-3, 1, 7, 0, None, 0, None), -1)
    expect(lib.fil_attn_fwd(*nul9, 4, 200, 64, 4, 16, 0.25, 1e-3, 1, 0, 24, None, 0, None), -1)
    expect(lib.fil_attn_fwd(*nul9, 4, 200, 16, 9, 16, 0.25, 1e-3, 1, 0, 0, None, 0, None), -4)
    expect(lib.fil_attn_fwd(*nul9, 4, 600, 16, 4, 16, 0.25, 1e-3, 1, 0, 0, None, 0, None), -4, b"512")
    expect(lib.fil_attn_bwd(*([None] * 17), 4, 200, 16, 4, 16, 0.25, 1e-3, 1, 0, 0, None, 0, None), -1)
    expect(lib.fil_score_add_sigmoid_fwd(None, None, None, None, None, 4, None), -1, b"bad argument")
    expect(lib.fil_score_add_sigmoid_fwd(None, None, None, None, None, 0, None), 0)
    expect(lib.fil_score_add_sigmoid_bwd(None, None, None, 4, None), -1)
    expect(lib.fil_bce_mean_fwd(None, None, 1e-7, None, None, 0, None), -1)
    expect(lib.fil_bce_mean_fwd(None, None, 0.7, None, None, 8, None), -1, b"eps")
    expect(lib.fil_gemm_f32(None, None, None, None, 4, 8, 16, 16, 8, 8, 0, 0, 0, None, 0, None), -1)           # NULL operands
    expect(lib.fil_gemm_f32(None, None, None, None, 0, 8, 16, 16, 8, 8, 0, 0, 0, None, 0, None), 0)            # empty
    expect(lib.fil_gemm_f32(None, None, None, None, 4, 8, 16, 16, 8, 8, 0, 0, 3, None, 0, None), -1)           # unknown epilogue
    assert lib.fil_gemm_f32_workspace_bytes(4096, 256, 637) == 0 and lib.fil_gemm_f32_workspace_bytes(4096, 128, 256) == 0 and lib.fil_gemm_f32_workspace_bytes(637, 256, 4096) > 0 and lib.fil_gemm_f32_workspace_bytes(637, 256, 4096) % (637 * 256 * 4) == 0
    expect(lib.fil_relu_bias_bwd(None, None, None, None, 4, 8, 0, None, 0, None), -1)
    expect(lib.fil_relu_bias_bwd(None, None, None, None, 4, 8, 5, None, 0, None), -1)
    assert lib.fil_relu_bias_bwd_workspace_bytes(4096, 256) == 256 + 64 * 256 * 4
    w2 = _lib.int_array([16, 128])
    pp = (ctypes.c_void_p * 2)(None, None)
    d2, dbad = _lib.int_array([0, 1]), _lib.int_array([0, 2])
    expect(lib.fil_merge_softmax_fwd(None, None, None, 2, None, None, None, 4, 2, None), -1, b"parts")
    expect(lib.fil_merge_softmax_fwd(pp, w2, d2, 2, None, None, None, 4, 9, None), -1, b"units")
    expect(lib.fil_merge_softmax_fwd(pp, w2, dbad, 2, None, None, None, 0, 2, None), -1, b"dtype")
    expect(lib.fil_merge_softmax_fwd(pp, w2, d2, 2, None, None, None, 0, 2, None), 0)
    expect(lib.fil_merge_softmax_fwd(pp, w2, d2, 2, None, None, None, 4, 2, None), -1)          # NULL parts with B > 0
    expect(lib.fil_merge_softmax_bwd(pp, w2, d2, 2, None, None, None, None, None, None, 4, 2, None, 0, None), -1)
    assert lib.fil_merge_softmax_bwd_workspace_bytes(4096, 144, 2) == 256 + 64 * 145 * 2 * 4
    assert lib.fil_merge_softmax_bwd_workspace_bytes(0, 144, 2) == 256 + 145 * 2 * 4
    small = lib.fil_attn_bwd_workspace_bytes(16, 200, 16, 4, 16, 1)
    assert 0 < small < 1 << 20 and lib.fil_attn_bwd_workspace_bytes(16, 200, 16, 4, 16, 0) >= small + 2 * 4 * 16 * 200 * 16 * 4
    assert lib.fil_attn_bwd_workspace_bytes(0, 200, 16, 4, 16, 1) == 0
    # product attention, embeddings
    expect(lib.fil_pattn_fwd(None, None, None, None, None, 4, 39, 39, 80, 8, 1.0, 0, None), -4)
    expect(lib.fil_pattn_fwd(None, None, None, None, None, 0, 39, 39, 8, 8, 1.0, 0, None), 0)
    expect(lib.fil_pattn_fwd(None, None, None, None, None, 4, 39, 39, 8, 8, 1.0, 0, None), -1)
    expect(lib.fil_pattn_bwd(None, None, None, None, None, None, None, None, 4, 0, 39, 8, 8, 1.0, 0, None), -1)
    expect(lib.fil_embed_gather(None, None, None, None, None, None, 4, 3, 8, None), -1)
    expect(lib.fil_embed_gather(None, None, None, None, None, None, 0, 3, 8, None), 0)
    expect(lib.fil_embed_scatter_add(None, None, None, None, None, 4, 3, 8, None), -1)
    expect(lib.fil_embed_row_ids(None, None, None, None, None, 4, 0, None), -1)
    expect(lib.fil_embed_sort_fields(None, None, None, None, None, None, 4, 0, 0, None), -1)
    expect(lib.fil_embed_sort_fields(None, None, None, None, None, None, 0, 3, 0, None), 0)
    one_ptr = ctypes.c_void_p(8)   # (never dereferenced: the size limit is checked before the launch)
    expect(lib.fil_embed_sort_fields(one_ptr, None, None, one_ptr, one_ptr, one_ptr, 9000, 3, 0, None), -4, b"8192")
    expect(lib.fil_embed_segment_sum(None, None, None, None, None, None, 5, 300, None), -4)
    expect(lib.fil_embed_run_sum(None, None, None, None, 0, 8, None), 0)
    expect(lib.fil_embed_run_sum_dt(None, None, None, None, 0, 8, 1, None), 0)
    expect(lib.fil_embed_run_sum_dt(None, None, None, None, 0, 8, 7, None), -1, b"g_dtype")
    expect(lib.fil_embed_gather_dt(None, None, None, None, None, None, 0, 3, 8, 1, None), 0)
    expect(lib.fil_embed_gather_dt(None, None, None, None, None, None, 4, 3, 8, 5, None), -1, b"out_dtype")
    expect(lib.fil_embed_gather_xt(None, None, None, None, None, None, None, 4, 3, 8, None), -1)
    expect(lib.fil_embed_gather_xt(None, None, None, None, None, None, None, 0, 3, 8, None), 0)
    one = ctypes.c_void_p(8)   # (never dereferenced: the LDS limit is checked before the launch)
    expect(lib.fil_embed_gather_xt(one, one, None, one, one, one, None, 4, 400, 64, None), -4, b"LDS")
    # the profiler's text protocol (no launches recorded: an empty table, correctly terminated, whatever the buffer size)
    assert lib.fil_profile_begin(b"cin_fwd_l,attn") == 0
    buf = ctypes.create_string_buffer(64)
    assert lib.fil_profile_end(buf, 64) == 1 and buf.value == b""
    assert lib.fil_profile_begin(None) == 0 and lib.fil_profile_end(None, 0) == 1
    # the error text of a long message is truncated, not overrun (fil_last_error is a 512-byte thread-local buffer)
    expect(lib.fil_cin_fwd(None, None, None, None, None, None, None, None, -1, 39, 16, 3, H3, 1, 0, None, 0, None), -1)
    assert len(lib.fil_last_error()) < 512
    return n


if __name__ == "__main__":
    print("host calls ok:", run(bind(sys.argv[1])))
