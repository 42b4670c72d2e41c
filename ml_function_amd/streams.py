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
_FALLBACK = {}      # same key -> [stream handed out after a failed probe round, failed rounds so far, time of the last one]
_MAX_FAILED_ROUNDS = 3      # probe rounds (of `tries` candidates each) before a key is served from _FALLBACK for good
_RETRY_AFTER_S = 30.0       # ... and the pause between two of them


def independent_stream(device=None, tries=8, priority=0, tag=None):
    """A stream on `device` that does not share a hardware queue with the current stream (the first of `tries` candidates that
    passes shares_queue_with_current; correctness never depends on it).  The probe drains the device (a >= 2 ms sleep kernel per
    candidate), so its result is kept per (device, current stream, tag): a caller that asks again -- every epoch's data iterator,
    every reducer -- gets its stream back; different tags get different streams.  When NO candidate passes (a single hardware
    queue, a serialising profiler, a loaded device) the last candidate is handed out and kept as the key's fallback: callers get
    that same stream back, and the probe is repeated at most _MAX_FAILED_ROUNDS times, _RETRY_AFTER_S apart (a passing candidate
    then replaces the fallback) -- not on every call, which would cost >= 16 ms of device drains each time.  Under stream capture
    no probe can run: the remembered stream, or a plain new one, is returned."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, priority, torch.cuda.current_stream(dev).cuda_stream, tag)
    if key in _INDEPENDENT:
        return _INDEPENDENT[key]
    fb = _FALLBACK.get(key)
    if torch.cuda.is_current_stream_capturing():
        return fb[0] if fb is not None else torch.cuda.Stream(device=dev, priority=priority)
    if fb is not None and (fb[1] >= _MAX_FAILED_ROUNDS or time.monotonic() - fb[2] < _RETRY_AFTER_S):
        return fb[0]
    cand = None
    with torch.cuda.device(dev):
        for _ in range(max(1, tries)):
            cand = torch.cuda.Stream(device=dev, priority=priority)
            if cand not in _INDEPENDENT.values() and not shares_queue_with_current(cand, dev):
                _INDEPENDENT[key] = cand      # only a candidate that PASSED is remembered as independent
                _FALLBACK.pop(key, None)
                return cand
    # no candidate passed: keep ONE fallback stream per key (callers that cache per stream see a stable object) and count the round
    if fb is None:
        _FALLBACK[key] = [cand, 1, time.monotonic()]
        return cand
    fb[1] += 1
    fb[2] = time.monotonic()
    return fb[0]
