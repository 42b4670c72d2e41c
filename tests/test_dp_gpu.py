"""The data-parallel reducer on a real GPU: a one-rank "nccl" (= RCCL) group with force=True runs the per-layer all-reduces of
bench.py's step -- side stream, waits on the library's grad_ready events, RCCL's kernels -- on the one GPU a test box has.
An all-reduce over one rank must leave every gradient bit for bit unchanged, whichever way it is scheduled."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_layerwise_allreduce_executes_on_one_rank():
    import torch.distributed as dist
    import bench
    from ml_function_amd import dp
    from ml_function_amd import functional as Fn
    device = torch.device("cuda", 0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with bench.stdout_to_stderr():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=device)
    try:
        inp = bench.make_inputs(0, device, batch=512)
        flat, grads, segments = bench.make_bucket(inp, device)
        L = len(inp["Ws"])

        def backward(ready=None):
            flat.zero_()
            out, pooled, saved = Fn.cin_forward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], inp["dense_b"], 1, 0)
            Fn.cin_backward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], pooled, saved, inp["g"], 1, 0, grads=grads,
                                ready_events=ready)

        backward()
        torch.cuda.synchronize()
        plain = flat.clone()
        assert float(plain.abs().max()) > 0

        calls = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(int(t.numel()))
            return real(t, *a, **k)

        dist.all_reduce = counting
        try:
            # (a) per layer on the side stream, keyed by the library's grad_ready events (bench.py's default)
            red = dp.LayerwiseAllReduce(flat, segments, force=True)
            assert red.active() and not dp.LayerwiseAllReduce(flat, segments).active()
            ready = [None] * (L + 1)
            for i in range(L):
                ready[L - 1 - i] = red.events[i]
            for _ in range(3):
                backward(ready)
                red.launch()
                red.wait()
            torch.cuda.synchronize()
            assert calls == [b - a for a, b in segments] * 3           # every segment really went through the backend
            assert torch.equal(flat, plain)
            # (b) one all-reduce of the whole bucket after the backward (bench.py --no-overlap)
            del calls[:]
            red1 = dp.LayerwiseAllReduce(flat, [(0, flat.numel())], force=True)
            backward()
            red1.events[0].record()
            red1.launch()
            red1.wait()
            torch.cuda.synchronize()
            assert calls == [flat.numel()] and torch.equal(flat, plain)
        finally:
            dist.all_reduce = real
    finally:
        dist.destroy_process_group()
