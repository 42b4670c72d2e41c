"""Does the K=64 backward run faster as two independent 4-wave workgroups per CU than as one 8-wave workgroup?  At F = 160 both forms fit
(x + k images + tiles + weight table = 79,872 bytes): the default picks WPH = 1 (two workgroups per CU, dx through global memory),
FIL_ATTN_WPH=2 forces the one-workgroup form (dx image in LDS).  Run on the GPU box:  python tools/attn_occ_probe.py [F] [K_in]
(each setting in a child process: the knobs are read once)."""
import os
import subprocess
import sys

F = int(sys.argv[1]) if len(sys.argv) > 1 else 160
KIN = int(sys.argv[2]) if len(sys.argv) > 2 else 64
CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
from ml_function_amd import _lib, synth
from ml_function_amd import functional as Fn
F, KIN = int(sys.argv[1]), int(sys.argv[2])
c = synth.attn_stack_case(4096, F, KIN, 4, 16, 1)
t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
x = t(c["x"]).requires_grad_()
layers = [tuple(t(p).requires_grad_() for p in lay) for lay in c["layers"]]
dy = t(c["dy"])
for _ in range(5):
    Fn.autoint_stack(x, layers, precision="f16_mfma").backward(dy)
torch.cuda.synchronize()
_lib.profile_begin(None)
for _ in range(20):
    Fn.autoint_stack(x, layers, precision="f16_mfma").backward(dy)
torch.cuda.synchronize()
p = _lib.profile_end()
print({k: round(v["avg_ms"], 4) for k, v in p.items()})
'''
for env in ({}, {"FIL_ATTN_WPH": "2"}, {"FIL_ATTN_WPH": "1", "FIL_ATTN_DX_LDS": "0"}):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", CHILD, str(F), str(KIN)], env=e, capture_output=True, text=True)
    print("F=%d K_in=%d %s:" % (F, KIN, env or "default"), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
