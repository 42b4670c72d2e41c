#!/usr/bin/env python
"""Where does the fixed cost of the per-layer collectives go?  One GPU, one-rank RCCL group, the benchmark's CIN step with:
  a  nothing (plain step)                      b  the library's grad_ready events recorded, nobody waits
  c  + the side stream waits on each event     d  + the all-reduces on the side stream (LayerwiseAllReduce.launch)
  e  + the compute stream joins the side stream (LayerwiseAllReduce.wait): the full data-parallel step
Prints ms per step of each (median of 5 x 20 steps) and the host time to enqueue one step."""
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ml_function_amd import dp  # noqa: E402
from ml_function_amd import functional as Fn  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with bench.stdout_to_stderr():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        dist.barrier()
    inp = bench.make_inputs(0, dev)
    flat, grads, segments = bench.make_bucket(inp, dev)
    red = dp.LayerwiseAllReduce(flat, segments, force=True)
    L = len(inp["Ws"])
    ready = [None] * (L + 1)
    for i in range(L):
        ready[L - 1 - i] = red.events[i]

    def compute(ev):
        out, pooled, saved = Fn.cin_forward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], inp["dense_b"], 1, 0)
        Fn.cin_backward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], pooled, saved, inp["g"], 1, 0, grads=grads, ready_events=ev)

    def variant(name):
        def step():
            if name == "a":
                compute(None)
            elif name == "b":
                compute(ready)
            elif name == "c":
                compute(ready)
                with torch.cuda.stream(red.side):
                    for ev in red.events:
                        red.side.wait_event(ev)
            elif name == "d":
                compute(ready)
                red.launch()
                red._works = []
            elif name == "f":      # the all-reduces without the side stream's wait on the backend's stream
                compute(ready)
                with torch.cuda.stream(red.side):
                    for ev, (lo, hi) in zip(red.events, red.segments):
                        red.side.wait_event(ev)
                        dist.all_reduce(flat[lo:hi], async_op=True)
            elif name == "g":      # three all-reduces on the compute stream after the backward
                compute(None)
                for lo, hi in red.segments:
                    dist.all_reduce(flat[lo:hi])
            elif name == "h":      # one all-reduce on the compute stream after the backward
                compute(None)
                dist.all_reduce(flat)
            else:
                compute(ready)
                red.launch()
                red.wait()
        for _ in range(5):
            step()
        ts, hs = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
            hs.append((t1 - t0) / 20 * 1e3)
        return float(np.median(ts)), float(np.median(hs))

    def aliases_main(stream):
        """Does `stream` share a hardware queue with the current stream?  A tiny fill on it cannot finish while a long kernel
        occupies the current stream if (and only if) both feed the same in-order queue."""
        probe = torch.zeros(64, device=dev)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(4e6))          # ~2 ms on the compute stream
        done = torch.cuda.Event()
        with torch.cuda.stream(stream):
            probe.fill_(1.0)
            done.record()
        t0 = time.perf_counter()
        while not done.query() and time.perf_counter() - t0 < 0.0008:
            pass
        early = done.query()
        torch.cuda.synchronize()
        return not early

    cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
    al = [aliases_main(c) for c in cands]
    print("side-stream candidates sharing the compute stream's hardware queue:", al, " reducer's own:", aliases_main(red.side), flush=True)
    for want in (False, True):
        pick = [c for c, a in zip(cands, al) if a == want]
        if pick:
            red.side = pick[0]
            for name in ("c", "d", "e"):
                ms, host = variant(name)
                print("side stream %s: %s  %.4f ms" % ("ALIASED with" if want else "independent of", name, ms), flush=True)
    for name, what in (("a", "plain step"), ("b", "+ grad_ready events recorded"), ("c", "+ side stream waits on them"),
                       ("d", "+ all-reduces on the side stream"), ("e", "+ compute stream joins the side stream"),
                       ("f", "d without work.wait() on the side stream"), ("g", "three all-reduces on the compute stream"),
                       ("h", "one all-reduce on the compute stream"), ("a", "plain step again")):
        ms, host = variant(name)
        print("%s  %-42s %.4f ms   (host enqueue %.4f ms)" % (name, what, ms, host), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
