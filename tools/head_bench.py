"""Times the model-head kernels in isolation (in-library HIP events): python tools/head_bench.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_function_amd import _lib
from ml_function_amd import functional as Fn

B = 4096
for dt in (torch.float32, torch.bfloat16):
    a = torch.randn(B, 16, device="cuda").to(dt).requires_grad_()
    b = torch.randn(B, 128, device="cuda").to(dt).requires_grad_()
    W = torch.randn(144, 2, device="cuda").requires_grad_()
    bias = torch.zeros(2, device="cuda").requires_grad_()
    g = torch.randn(B, 2, device="cuda")
    x = torch.randn(B, 637, device="cuda").requires_grad_()
    W1 = torch.randn(637, 256, device="cuda").requires_grad_()
    b1 = torch.zeros(256, device="cuda").requires_grad_()
    g1 = torch.randn(B, 256, device="cuda")
    def step():
        Fn.merge_softmax([a, b], W, bias).backward(g)
        if dt == torch.bfloat16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = Fn.dense_relu(x, W1, b1)
        else:
            y = Fn.dense_relu(x, W1, b1)
        y.backward(g1.to(y.dtype))
    for _ in range(5):
        step()
    _lib.profile_begin(None)
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    print(dt, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.profile_end().items()})

# the dense layers' GEMMs: library kernel vs torch (hipBLASLt) at the xDeepFM MLP's shapes
import time
def tm(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K, ta, tb) in [(4096, 256, 637, 0, 0), (4096, 637, 256, 0, 1), (637, 256, 4096, 1, 0), (4096, 128, 256, 0, 0), (4096, 256, 128, 0, 1),
                          (256, 128, 4096, 1, 0), (4096, 64, 128, 0, 0), (128, 64, 4096, 1, 0)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda"); b = torch.randn((N, K) if tb else (K, N), device="cuda")
    t_fil = tm(lambda: Fn.gemm_f32(a, b, trans_a=bool(ta), trans_b=bool(tb)))
    _lib.profile_begin("gemm_f32")
    for _ in range(20): Fn.gemm_f32(a, b, trans_a=bool(ta), trans_b=bool(tb))
    torch.cuda.synchronize()
    kern = _lib.profile_end()["gemm_f32"]["avg_ms"] * 1e3
    t_ref = tm(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b))
    print("gemm M=%d N=%d K=%d ta=%d tb=%d: fil %.1f us host-timed, %.1f us in events (%.1f TFLOP/s)  torch %.1f us host-timed" % (M, N, K, ta, tb, t_fil, kern, 2.0 * M * N * K / kern / 1e6, t_ref))
