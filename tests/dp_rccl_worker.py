"""One rank of the N-rank RCCL check (TEST INFRASTRUCTURE; launched by tests/test_dp_gpu.py through torch.distributed.run, one process per
GPU): every rank runs the CIN backward of ITS row shard through the HIP path with the layer-wise all-reduce overlapped behind the
library's grad_ready events (dp.LayerwiseAllReduce, exactly bench.py's step), then rank 0 compares the reduced bucket with the
gradients of the FULL batch computed on its own GPU: shard gradients add up to the full-batch gradient (fp32 reassociation), and every
rank holds bit-identical reduced gradients.  Prints 'DP_RCCL_OK <world> <max rel err>' on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from ml_function_amd import dp
    from ml_function_amd import functional as Fn
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    with bench.stdout_to_stderr():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        dist.barrier()
    per_rank = int(os.environ.get("FIL_DP_ROWS", "1536"))      # 1536 x 16 = 24,576 rows per rank: above the quadratic tail's size rule
    inp = bench.make_inputs(rank, device, batch=per_rank, param_batch=per_rank)
    flat, grads, segments = bench.make_bucket(inp, device)
    L = len(inp["Ws"])
    sh0 = inp["x"].shape
    points = Fn.cin_grad_ready_points(sh0[0], sh0[1], sh0[2], [int(w.shape[1]) for w in inp["Ws"]], 0)
    points = dp.agree_on_points(points, device)
    merged, layer_of_event = dp.merge_segments_by_point(segments, points)
    reducer = dp.LayerwiseAllReduce(flat, merged, force=(world == 1))   # (one rank: the collectives are issued anyway, plumbing only)
    assert reducer.active()
    ready = [None] * (L + 1)
    for ev, l in zip(reducer.events, layer_of_event):
        ready[l] = ev

    def backward(x, g, target, ready_events):
        out, pooled, saved = Fn.cin_forward_raw(x, inp["Ws"], inp["bs"], inp["dense_w"], inp["dense_b"], 1, 0)
        Fn.cin_backward_raw(x, inp["Ws"], inp["bs"], inp["dense_w"], pooled, saved, g, 1, 0, grads=target, ready_events=ready_events)

    for _ in range(3):          # (repeats: the side stream, the events and the collectives are reused as in a training loop)
        backward(inp["x"], inp["g"], grads, ready)
        reducer.launch()
        reducer.wait()
    torch.cuda.synchronize()
    reduced = flat.clone()
    # every rank holds the same reduced bucket, bit for bit
    every = [torch.empty_like(reduced) for _ in range(world)]
    dist.all_gather(every, reduced)
    same = all(torch.equal(e, every[0]) for e in every)
    # the full batch on rank 0's own GPU: all shards' inputs gathered, one backward, no collective
    xs = [torch.empty_like(inp["x"]) for _ in range(world)]
    gs = [torch.empty_like(inp["g"]) for _ in range(world)]
    dist.all_gather(xs, inp["x"])
    dist.all_gather(gs, inp["g"])
    ok, err = True, 0.0
    if rank == 0:
        full_flat, full_grads, _ = bench.make_bucket(inp, device)
        full_grads["dx"] = torch.empty((world * per_rank,) + tuple(sh0[1:]), dtype=torch.float32, device=device)
        backward(torch.cat(xs), torch.cat(gs), full_grads, None)
        torch.cuda.synchronize()
        err = float((reduced.double() - full_flat.double()).abs().max() / full_flat.double().abs().max())
        ok = same and err < 1e-5
        print("DP_RCCL_%s %d %.3e" % ("OK" if ok else "FAILED", world, err), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
