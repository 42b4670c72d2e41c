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
            # (a') bench.py's default: one collective per point of the backward (fused tail: the two top layers together)
            del calls[:]
            H = [int(w.shape[1]) for w in inp["Ws"]]
            points = Fn.cin_grad_ready_points(512, inp["x"].shape[1], inp["x"].shape[2], H, 0)
            assert points == [2, 1, 1, 0]                   # layer 0 last; layers 1 and 2 (the fused tail) together; head first
            assert Fn.cin_grad_ready_points(512, inp["x"].shape[1], inp["x"].shape[2], H, 32) == [3, 2, 1, 0]   # FIL_CIN_NOTAIL
            merged, layer_of_event = dp.merge_segments_by_point(segments, points)
            assert merged == [(segments[0][0], segments[1][1]), segments[2]] and layer_of_event == [1, 0]
            redm = dp.LayerwiseAllReduce(flat, merged, force=True)
            ready = [None] * (L + 1)
            for ev, l in zip(redm.events, layer_of_event):
                ready[l] = ev
            for _ in range(2):
                backward(ready)
                redm.launch()
                redm.wait()
            torch.cuda.synchronize()
            assert calls == [b - a for a, b in merged] * 2 and torch.equal(flat, plain)
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


def test_independent_stream_runs_beside_the_compute_stream():
    """torch's streams are multiplexed onto a few in-order hardware queues; the reducer's (and the input pipeline's) side stream
    must be one whose work proceeds while the compute stream is busy."""
    from ml_function_amd import streams
    device = torch.device("cuda", 0)
    side = streams.independent_stream(device)
    assert not streams.shares_queue_with_current(side, device)
    assert streams.shares_queue_with_current(torch.cuda.current_stream(device), device)     # the probe itself: same queue = True
    t = torch.zeros(1 << 20, device=device)
    with torch.cuda.stream(side):
        t.add_(1.0)                                   # (first launch of this kernel: code-object load, not what is timed)
    torch.cuda.synchronize()
    start, busy, done = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    start.record()
    torch.cuda._sleep(int(2e7))
    busy.record()
    with torch.cuda.stream(side):
        t.add_(1.0)
        done.record()
    torch.cuda.synchronize()
    assert start.elapsed_time(done) < 0.5 * start.elapsed_time(busy)      # the side work finished under the long compute kernel
    assert float(t.sum()) == float(2 << 20)


def test_bench_line_accounting():
    """bench.py's derived fields against each other (the dominant scope's record covers W + K launches, and dividing
    it by K printed 1.25 launches per step and an executed fraction 8 % high): every GEMM scope runs once per step, the step's
    executed flops are the three merged GEMM launches (2 * 65,536 rows * 780 pairs * 256 columns each, + 1 % for the shortcut columns),
    and executed_frac / roofline.frac follow from the printed times."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "3", "--windows", "2", "--no-side",
                        "--no-cpu-baseline", "--no-graph-replay"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    gemms = {k: v for k, v in res["kernels"].items() if v.get("executed_flops_per_launch", 0) > 1e9}    # (cin_fwd_prep shares the prefix: 20 MFLOP)
    assert len(gemms) == 3 and all(abs(v["launches_per_step"] - 1.0) < 1e-9 for v in gemms.values()), gemms
    per_gemm = 2.0 * 65536 * 780 * 256
    assert 3 * per_gemm <= res["executed_flops_per_step"] <= 3 * per_gemm * 1.01
    assert res["executed_flops_per_step"] == pytest.approx(sum(v["executed_flops_per_launch"] for v in gemms.values()), rel=1e-3)
    assert res["executed_frac"] == pytest.approx(res["executed_flops_per_step"] / (res["ms_per_step"] * 1e-3) / (res["roofline"]["peak"] * 1e12), rel=1e-9)
    rf = res["roofline"]
    assert rf["frac"] == pytest.approx(rf["flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12 / rf["peak"], rel=1e-9)
    assert rf["avg_launch_ms"] < res["ms_per_step"] and res["executed_frac"] < rf["frac"] + 0.1
    assert res["value"] == pytest.approx(4096 / (res["ms_per_step"] * 1e-3), rel=1e-9)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(script_args, n, timeout=900):
    """n fresh ranks through torch.distributed.run (children of this process; nothing here hands them a GPU context)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)


def test_rccl_worker_on_one_rank():
    """The N-rank RCCL check of tests/dp_rccl_worker.py at world size 1 (what a one-GPU box can run): the worker itself, its launcher and
    the overlapped reducer; the comparison is then the shard against itself."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks([os.path.join(root, "tests", "dp_rccl_worker.py")], 1)
    assert r.returncode == 0 and "DP_RCCL_OK 1" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node; a gpurun box has one)")
def test_two_ranks_over_rccl():
    """World size 2 over RCCL / xGMI (skipped on a one-GPU box): (a) the sum of the two shards' gradients after the overlapped
    layer-wise all-reduce == the full-batch gradient, identical on both ranks (tests/dp_rccl_worker.py); (b) bench.py --gpus 2 prints
    its line with both ranks seen and the collectives' fields filled."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks([os.path.join(root, "tests", "dp_rccl_worker.py")], 2)
    assert r.returncode == 0 and "DP_RCCL_OK 2" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    r = _run_ranks([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "1", "--no-cpu-baseline"], 2)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["rccl"]["world_size_seen"] == 2 and res["config"]["global_batch"] == 2 * 4096
    assert res["value"] == pytest.approx(2 * 4096 / (res["ms_per_step"] * 1e-3), rel=1e-6)
    assert len(res["rccl"]["ms_per_step_per_rank"]) == 2 and res["rccl"]["allreduce_alone_ms"] > 0
