"""embed gradient kernels (sort + run sum) against the id distribution: bench vocabularies (10 ... 1e5, log-uniform) vs all-large.
usage (GPU box): python tools/embed_bench.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ml_function_amd import _lib, models  # noqa: E402
from ml_function_amd.layers.interactive_layer import SparseEmbed  # noqa: E402,F401

dev = torch.device("cuda", 0)
B = 4096
for name, lo in (("bench vocab 10..1e5", 10), ("vocab 1e4..1e5", 1e4), ("vocab 100..1e5", 100)):
    rng = np.random.default_rng(2020)
    vocab = [int(v) for v in np.exp(rng.uniform(np.log(lo), np.log(1e5), 39))]
    fi = models.FeatureInput(sparseInfo=models.make_sparse_info(vocab, embed_dim=16), useLinear=True)
    body = models.FM()
    model = models.CTRModel(fi, body).to(dev)
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in vocab], 1), device=dev)
    out = model(None, idx)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        model(None, idx).sum().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if "embed" in e.name:
            rows.setdefault(e.name[:40], []).append(e.device_time)
    print(name, {k: (len(v) // 10, round(sum(v) / len(v), 1)) for k, v in rows.items()})
