"""Host-side cost of the per-call plumbing around a C-ABI call (microseconds per call, median of 5 x 2000)."""
import time
import torch

x = torch.empty(1024, device="cuda")
def t(fn, n=2000):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
print("is_current_stream_capturing   %.2f us" % t(torch.cuda.is_current_stream_capturing))
print("current_stream().cuda_stream  %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("_cuda_getCurrentRawStream     %.2f us" % t(lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())))
print("torch.empty(1 MB uint8)       %.2f us" % t(lambda: torch.empty(1 << 20, dtype=torch.uint8, device="cuda")))
print("torch.empty_like              %.2f us" % t(lambda: torch.empty_like(x)))
