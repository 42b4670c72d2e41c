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
