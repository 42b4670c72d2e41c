"""Which call form of the MLP's bf16 weight-gradient GEMM (dW = x^T dz, reduction over the batch) does the library run fastest?
usage (GPU box): python tools/mm_forms.py"""
import torch

dev = torch.device("cuda", 0)
B = 4096


def timeit(fn, n=20):
    """us per call, replayed from a HIP graph (eagerly the host's ~18 us per call hides everything)"""
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


for (I, O) in ((637, 256), (256, 128), (128, 64), (640, 256)):
    x = torch.randn(B, I, device=dev, dtype=torch.bfloat16)
    dz = torch.randn(B, O, device=dev, dtype=torch.bfloat16)
    xt, dzt = x.t().contiguous(), dz.t().contiguous()
    forms = {
        "x.t() @ dz": lambda: x.t() @ dz,
        "(dz.t() @ x).t()": lambda: (dz.t() @ x),
        "xt_c @ dz": lambda: xt @ dz,
        "dzt_c @ x": lambda: dzt @ x,
        "f32 x.t() @ dz": lambda: x.float().t() @ dz.float(),
        "split8 bmm": lambda: torch.bmm(x.view(8, B // 8, I).transpose(1, 2), dz.view(8, B // 8, O)).sum(0),
        "split16 bmm": lambda: torch.bmm(x.view(16, B // 16, I).transpose(1, 2), dz.view(16, B // 16, O)).sum(0),
        "split32 bmm f32out": lambda: torch.bmm(x.view(32, B // 32, I).transpose(1, 2), dz.view(32, B // 32, O)).float().sum(0),
    }
    print("I=%d O=%d" % (I, O), "  ".join("%s: %.1f" % (k, timeit(f)) for k, f in forms.items()))
