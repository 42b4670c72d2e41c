#!/usr/bin/env python
"""Per-gradient error table of the AutoInt stack (config 5 layer shapes) against the fp64 oracle, f32 and f16-MFMA modes, on
the kinked and the kink-free inputs: what the bars of tests/test_golden.py / test_gpu_parity.py are set from."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_function_amd import functional as Fn  # noqa: E402
from ml_function_amd import synth  # noqa: E402
from oracle import closed  # noqa: E402


def rel(a, b):
    a = a.detach().double().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rms(a, b):
    a = a.detach().double().cpu().numpy()
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-300))


def main():
    dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    for B in (2, 32):
        for shift, center in ((0.0, False), (4.0, True), (4.0, False)):
            c = synth.attn_stack_case(B, 200, 16, 4, 16, 3, dist="normal", beta_shift=shift, center_upper=center)
            y_o = closed.attn_stack_fwd(c["x"], c["layers"])
            dx_o, g_o = closed.attn_stack_bwd(c["x"], c["layers"], c["dy"])
            for prec in ("f32", "f16_mfma"):
                x = dev(c["x"]).requires_grad_()
                layers = [tuple(dev(p).requires_grad_() for p in lay) for lay in c["layers"]]
                y = Fn.autoint_stack(x, layers, precision=prec)
                y.backward(dev(c["dy"]))
                row = ["B=%d shift=%g center=%d %-8s y %.1e dx %.1e" % (B, shift, center, prec, rel(y, y_o), rel(x.grad, dx_o))]
                for l in range(3):
                    row.append(" | L%d " % l + " ".join("%s %.1e/%.1e" % (n, rel(p.grad, w), rms(p.grad, w))
                                                        for p, w, n in zip(layers[l], g_o[l], ["Wq", "Wk", "Wr", "g", "b"])))
                print("".join(row), flush=True)


if __name__ == "__main__":
    main()
