"""HIP streams that really run beside the compute stream.

HIP multiplexes its streams onto a handful of in-order hardware queues (four per device by default): every fourth stream
torch hands out feeds the SAME queue as the current stream.  Work on such a stream is not concurrent with the compute stream
at all -- a copy or a collective "on the side" waits behind every kernel enqueued before it, and each wait it carries stalls
the compute kernels behind it (measured on MI355X: the per-layer gradient all-reduce machinery cost 0.13 ms per 1.22 ms CIN
step on an aliasing side stream, 0.03 ms on an independent one; tools/rccl_overhead.py).  There is no API to ask which
queue a stream feeds, so independent_stream() finds out by experiment.
"""
import time

import torch


def shares_queue_with_current(stream, device=None):
    """True if a tiny fill on `stream` cannot finish while a ~2 ms kernel occupies the current stream, i.e. both feed the same
    in-order hardware queue.  Synchronises the device (call it at set-up time, never inside a graph capture)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    probe = torch.zeros(64, device=dev)
    torch.cuda.synchronize(dev)
    busy = torch.cuda.Event()
    torch.cuda._sleep(int(5e6))                      # >= 2 ms at <= 2.4 GHz
    busy.record()
    done = torch.cuda.Event()
    with torch.cuda.stream(stream):
        probe.fill_(1.0)
        done.record()
    t0 = time.perf_counter()
    while not done.query() and not busy.query() and time.perf_counter() - t0 < 0.05:
        pass
    alone = done.query() and not busy.query()        # finished while the compute stream was still busy
    torch.cuda.synchronize(dev)
    return not alone


_INDEPENDENT = {}   # (device index, priority, current stream, tag) -> the stream found by the probe: the answer cannot change


def independent_stream(device=None, tries=8, priority=0, tag=None):
    """A stream on `device` that does not share a hardware queue with the current stream (the first of `tries` candidates that
    passes shares_queue_with_current; the last candidate, NOT remembered, if none does -- correctness never depends on it).  The probe drains the
    device (a >= 2 ms sleep kernel per candidate), so its result is kept per (device, current stream, tag): a caller that asks again
    -- every epoch's data iterator, every reducer -- gets its stream back; different tags get different streams.  Under stream
    capture no probe can run: a plain new stream is returned."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, priority, torch.cuda.current_stream(dev).cuda_stream, tag)
    if key in _INDEPENDENT:
        return _INDEPENDENT[key]
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=dev, priority=priority)
    cand = None
    with torch.cuda.device(dev):
        for _ in range(max(1, tries)):
            cand = torch.cuda.Stream(device=dev, priority=priority)
            if cand not in _INDEPENDENT.values() and not shares_queue_with_current(cand, dev):
                _INDEPENDENT[key] = cand      # only a candidate that PASSED is remembered
                return cand
    # no candidate passed (a loaded device, a flaky probe): the last one is handed out uncached, so the next caller probes again
    # instead of being pinned to a stream that may serialise with the compute stream for the rest of the process
    return cand
