"""CPU / gloo stand-ins for the HIP entry points bench.py calls (TEST INFRASTRUCTURE, selected with `bench.py --stub
tests.bench_stub`): the launcher, the layer-wise all-reduce and the step logic of bench.py run unchanged on 2 CPU ranks,
with the gradients of a tiny CIN coming from the oracle instead of the kernels."""
import os

import numpy as np
import torch

from oracle import closed

SHAPE = dict(batch=12, conv=[6, 7], fields=5, embed=4)


class StubFn:
    @staticmethod
    def cin_forward_raw(x, Ws, bs, dense_w, dense_b, output_dim=1, mode=0):
        n = lambda t: t.detach().numpy()
        out, _, pooled = closed.cin_fwd(n(x), [n(w) for w in Ws], [n(b) for b in bs], n(dense_w), n(dense_b), return_maps=True)
        return torch.tensor(out, dtype=torch.float32), torch.tensor(pooled, dtype=torch.float32), None

    @staticmethod
    def cin_backward_raw(x, Ws, bs, dense_w, pooled, saved, g, output_dim=1, mode=0, grads=None, ready_events=None):
        assert ready_events is None          # no streams on the CPU
        n = lambda t: t.detach().numpy()
        dx, dWs, dbs, ddw, ddb = closed.cin_bwd(n(x), [n(w) for w in Ws], [n(b) for b in bs], n(dense_w), n(g)[:, None])
        grads["dx"].copy_(torch.tensor(dx, dtype=torch.float32))
        for l in range(len(Ws)):
            grads["dW"][l].copy_(torch.tensor(dWs[l], dtype=torch.float32))
            grads["db"][l].copy_(torch.tensor(dbs[l], dtype=torch.float32))
        grads["ddw"].copy_(torch.tensor(ddw, dtype=torch.float32))
        grads["ddb"].copy_(torch.tensor(ddb, dtype=torch.float32))
        return grads


def _on_done(flat, rank, world):
    out = os.environ.get("FIL_STUB_OUT")
    if out:
        np.save(os.path.join(out, "flat%d.npy" % rank), flat.numpy())


def install(ns):
    shape = dict(SHAPE)
    if os.environ.get("FIL_STUB_BATCH"):       # tests: a global batch that the ranks cannot share evenly
        shape["batch"] = int(os.environ["FIL_STUB_BATCH"])
    ns.update(backend="gloo", device=torch.device("cpu"), Fn=StubFn, profile=False, shape=shape, on_done=_on_done)
