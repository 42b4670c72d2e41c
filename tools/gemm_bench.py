"""fil_gemm_f32 at the xDeepFM MLP's shapes (forward with bias + ReLU, dx = dz W^T, dW = x^T dz), us per call replayed from a HIP graph.
usage (GPU box): [FIL_GEMM_KW=1] python tools/gemm_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from ml_function_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096


def timeit(fn, n=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * n) * 1e3


tot = 0.0
for (I, O) in ((637, 256), (256, 128), (128, 64), (64, 1)):
    x = torch.randn(B, I, device=dev)
    w = torch.randn(I, O, device=dev)
    b = torch.randn(O, device=dev)
    dz = torch.randn(B, O, device=dev)
    f = timeit(lambda: Fn.gemm_f32(x, w, bias=b, relu=True))
    dx = timeit(lambda: Fn.gemm_f32(dz, w, trans_b=True))
    dw = timeit(lambda: Fn.gemm_f32(x, dz, trans_a=True))
    tot += f + dx + dw
    print("I=%d O=%d  fwd %.1f  dx %.1f  dW %.1f us" % (I, O, f, dx, dw))
print("sum %.1f us" % tot)
