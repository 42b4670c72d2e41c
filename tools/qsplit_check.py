"""Split-bf16 mode of the merged quadratic tail (FIL_CIN_BF16X3) beside the exact mode at the benchmark shape: agreement of every
output / gradient, error of both against the fp64 graph oracle (B=512 shard), per-kernel times (in-library HIP events).
    python tools/qsplit_check.py [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ml_function_amd import _lib, synth  # noqa: E402
from ml_function_amd import functional as Fn  # noqa: E402


def run(c, mode, reps=1):
    dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    res = None
    for _ in range(reps):
        x = dev(c["x"]).requires_grad_()
        Ws = [dev(w).requires_grad_() for w in c["Ws"]]
        bs = [dev(b).requires_grad_() for b in c["bs"]]
        dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
        out = Fn.cin(x, Ws, bs, dw, db, mode=mode)
        out.backward(dev(c["g"]))
        res = dict(out=out.detach(), dx=x.grad, dW0=Ws[0].grad, dW1=Ws[1].grad, dW2=Ws[2].grad, db0=bs[0].grad, db1=bs[1].grad,
                   db2=bs[2].grad, ddw=dw.grad, ddb=db.grad)
    torch.cuda.synchronize()
    return res


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    c = synth.cin_case(B, 39, 16, [128, 128, 128])
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    r0 = run(c, 0)
    r2 = run(c, 2)
    print("mode 2 vs mode 0:", {k: "%.1e" % rel(r2[k], r0[k]) for k in r0})
    # both against the fp64 graph oracle on the first 512 samples (outputs / dx only: parameter gradients are sums over the batch)
    from oracle import graph
    n = min(B, 512)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    x = T(c["x"][:n]).requires_grad_()
    out = graph.cin(x, [T(w) for w in c["Ws"]], [T(b) for b in c["bs"]], T(c["dense_w"]), T(c["dense_b"]), output_dim=1)
    out.backward(T(c["g"][:n]))
    for name, r in (("mode 0", r0), ("mode 2", r2)):
        print(name, "vs fp64 oracle: out %.2e dx %.2e" % (rel(r["out"][:n], out.detach()), rel(r["dx"][:n], x.grad)))
    for mode in (0, 2):
        run(c, mode, reps=3)
        _lib.profile_begin(None)
        run(c, mode, reps=10)
        prof = _lib.profile_end()
        print("mode %d kernels (ms):" % mode, {k: round(v["avg_ms"], 4) for k, v in sorted(prof.items()) if k.startswith("cin_")},
              "sum %.4f" % sum(v["total_ms"] / 10 for k, v in prof.items() if k.startswith("cin_")))


if __name__ == "__main__":
    main()
