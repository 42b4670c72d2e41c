"""Diagnostic: where a wave of the fused attention backward spends its shader clocks (needs a -DFIL_ATTN_STAMPS build).
usage (on the GPU box):  FIL_HIPCC_FLAGS=-DFIL_ATTN_STAMPS python -m ml_function_amd.build --force && python tools/attn_stamps.py [layers] [precision] [K_in]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ml_function_amd import _lib, synth  # noqa: E402
from ml_function_amd import functional as Fn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "f16_mfma"
KIN = int(sys.argv[3]) if len(sys.argv) > 3 else 16      # 64 = the shape of layers 2.. of a stack (head-concat input), as a 1-layer case
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(1024 * 8 * 8, dtype=torch.int64, device="cuda")
lib.fil_attn_debug_stamps.argtypes = [ctypes.c_void_p]
lib.fil_attn_debug_stamps(buf.data_ptr())
c = synth.attn_stack_case(4096, 200, KIN, 4, 16, L)
t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
x = t(c["x"]).requires_grad_()
layers = [tuple(t(p).requires_grad_() for p in lay) for lay in c["layers"]]
names = ["0 stage+project_k", "1 block prologue (LN bwd, q proj)", "2 score tiles", "3 dq transposes + dW", "4 barrier wait",
         "5 cross-head dx", "6 last post-loop", "7 post-loop (dk image, dWk, dx += dk part)"]
for it in range(3):
    y = Fn.autoint_stack(x, layers, precision=prec)
    buf.zero_()
    y.backward(t(c["dy"]))      # the LAST layer's backward runs first; the buffer holds the stamps of the FIRST layer's (last launch)
    torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s.sum(1) > 0]
tot = s.sum(1)
print("waves %d  total cycles/wave mean %.0f (min %.0f max %.0f) -- layer with K_in=%d" % (len(s), tot.mean(), tot.min(), tot.max(), KIN))
import os
if os.environ.get("FIL_STAMPS_POST"):
    names = ["0 k projection (+ first inputs)", "1 the whole block loop", "2 dk merge (2 barriers)", "3 dWk", "4 dx += dk Wk^T", "5 barrier before the dx store",
             "6 x staging (+ the last sample's dx store)", "7 dx store + barrier at the top of the next sample"]
for i, n in enumerate(names):
    print("  %-45s %9.0f cycles  %5.1f %%" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean()))
