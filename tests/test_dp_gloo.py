"""Data-parallel path on CPU (gloo, world_size 2): shard bounds, the flat gradient bucket and its single
all-reduce.  Shard gradients come from the oracle (this is a test of the DP machinery, not of the kernels):
sum over ranks of grad(shard) must equal grad(full batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ml_function_amd import dp, synth
from oracle import closed


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bucket_views_alias_flat():
    b = dp.GradBucket([(3, 2), (4,), (1,)], torch.device("cpu"))
    assert b.flat.numel() == 11 and b.nbytes() == 44
    b.views[1].fill_(2.0)
    assert float(b.flat.sum()) == 8.0 and b.views[0].shape == (3, 2)
    assert b.views[2].data_ptr() == b.flat.data_ptr() + 10 * 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, F, K, conv = 10, 5, 4, [6, 7]
        c = synth.cin_case(B, F, K, conv, dist="normal")
        lo, hi = dp.shard_bounds(B, rank, world)
        assert tuple(dp.shard_rows(torch.tensor(c["x"])).shape) == (hi - lo, F, K)
        dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"][lo:hi], c["Ws"], c["bs"], c["dense_w"], c["g"][lo:hi])
        grads = list(dWs) + list(dbs) + [ddw, ddb]
        bucket = dp.GradBucket([g.shape for g in grads], torch.device("cpu"))
        for v, g in zip(bucket.views, grads):
            v.copy_(torch.tensor(g, dtype=torch.float32))
        bucket.all_reduce()
        # module-level helper: same result through parameter .grad fields
        lin = torch.nn.Linear(3, 2)
        for p in lin.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        dp.allreduce_module_grads(lin)
        assert all(float(p.grad.flatten()[0]) == 3.0 for p in lin.parameters())
        if rank == 0:
            np.save(os.path.join(out_dir, "flat.npy"), bucket.flat.numpy())
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    flat = np.load(tmp_path / "flat.npy")
    B, F, K, conv = 10, 5, 4, [6, 7]
    c = synth.cin_case(B, F, K, conv, dist="normal")
    _, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"])
    want = np.concatenate([g.reshape(-1) for g in list(dWs) + list(dbs) + [ddw, ddb]])
    assert flat.shape == want.shape
    assert np.abs(flat - want).max() / np.abs(want).max() < 1e-6


def _layerwise_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, F, K, conv = 10, 5, 4, [6, 7, 3]
        c = synth.cin_case(B, F, K, conv, dist="normal")
        lo, hi = dp.shard_bounds(B, rank, world)
        _, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"][lo:hi], c["Ws"], c["bs"], c["dense_w"], c["g"][lo:hi])
        sizes, segments, index = dp.cin_bucket_layout([w.shape for w in dWs], [b.shape for b in dbs], [ddw.shape, (1,)])
        flat = torch.zeros(sum(sizes))
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        put = lambda key, a: flat[offs[index[key]]:offs[index[key] + 1]].copy_(torch.tensor(np.asarray(a, np.float32).reshape(-1)))
        for l in range(len(conv)):
            put(("W", l), dWs[l])
            put(("b", l), dbs[l])
        put(("head", 0), ddw)
        put(("head", 1), ddb)
        red = dp.LayerwiseAllReduce(flat, segments)
        assert red.active() and red.events == [None] * len(conv)
        red.launch()
        red.wait()
        if rank == 0:
            np.save(os.path.join(out_dir, "flat.npy"), flat.numpy())
    finally:
        dist.destroy_process_group()


def test_cin_bucket_layout_is_in_readiness_order():
    sizes, segments, index = dp.cin_bucket_layout([(15, 6), (30, 7), (35, 3)], [(6,), (7,), (3,)], [(12, 1), (1,)])
    # [head | layer 2 | layer 1 | layer 0]; segment 0 (head + top layer) is final first, then one layer at a time downwards
    assert sizes == [12, 1, 105, 3, 210, 7, 90, 6]
    assert segments == [(0, 121), (121, 338), (338, 434)] and segments[-1][1] == sum(sizes)
    assert index[("W", 2)] == 2 and index[("b", 0)] == 7 and index[("head", 1)] == 1


def test_segments_merge_where_gradients_finish_together():
    sizes, segments, index = dp.cin_bucket_layout([(15, 6), (30, 7), (35, 3)], [(6,), (7,), (3,)], [(12, 1), (1,)])
    # every layer at its own point (no fused tail): nothing merges, events in layer order L-1 .. 0
    assert dp.merge_segments_by_point(segments, [3, 2, 1, 0]) == (segments, [2, 1, 0])
    # fused tail: layers 2 and 1 become final together -> one collective behind layer 1's slot, then layer 0
    assert dp.merge_segments_by_point(segments, [2, 1, 1, 0]) == ([(0, 338), (338, 434)], [1, 0])
    # everything at once (empty batch)
    assert dp.merge_segments_by_point(segments, [0, 0, 0, 0]) == ([(0, 434)], [0])
    # merged quadratic tail: layer 0 is final BEFORE layers 1 + 2 -> its segment is issued first
    assert dp.merge_segments_by_point(segments, [0, 1, 1, 0]) == ([(338, 434), (0, 338)], [0, 1])
    # four layers with the tail on top
    segs4 = [(0, 10), (10, 30), (30, 60), (60, 100)]
    assert dp.merge_segments_by_point(segs4, [3, 2, 1, 1, 0]) == ([(0, 30), (30, 60), (60, 100)], [2, 1, 0])


def _points_worker(rank, world, port, out_dir):
    """Ranks whose LOCAL batch sits on different sides of the library's tail threshold report different readiness points
    (global 4097 rows, K=16, 4 ranks: one shard of 1025 rows -> [0,1,1,0], three of 1024 -> [2,1,1,0]).  Every rank must still
    issue the same collectives in the same order: agree_on_points -> None -> per-layer segments in bucket order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes, segments, index = dp.cin_bucket_layout([(15, 6), (30, 7), (35, 3)], [(6,), (7,), (3,)], [(12, 1), (1,)])
        mine = [0, 1, 1, 0] if rank == 0 else [2, 1, 1, 0]
        agreed = dp.agree_on_points(mine)
        assert agreed is None
        segs, layer_of_event = dp.merge_segments_by_point(segments, agreed)
        assert segs == segments and layer_of_event == [2, 1, 0]          # the same on both ranks whatever `mine` was
        # and the reduction through them is the plain sum
        flat = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1)
        red = dp.LayerwiseAllReduce(flat, segs)
        red.launch()
        red.wait()
        assert torch.equal(flat, torch.arange(sum(sizes), dtype=torch.float32) * 3)
        # identical vectors pass through unchanged
        assert dp.agree_on_points([0, 1, 1, 0]) == [0, 1, 1, 0]
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_ranks_with_different_readiness_points_issue_the_same_collectives(tmp_path):
    mp.spawn(_points_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
    # one rank: nothing to agree on
    assert dp.agree_on_points([2, 1, 1, 0]) == [2, 1, 1, 0]


def test_layerwise_allreduce_equals_full_batch(tmp_path):
    """The bucket reduced segment by segment (the order bench.py overlaps them in) == gradient of the full batch."""
    world = 2
    mp.spawn(_layerwise_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    flat = np.load(tmp_path / "flat.npy")
    B, F, K, conv = 10, 5, 4, [6, 7, 3]
    c = synth.cin_case(B, F, K, conv, dist="normal")
    _, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"])
    want = np.concatenate([ddw.reshape(-1), ddb.reshape(-1)] + [np.concatenate([dWs[l].reshape(-1), dbs[l].reshape(-1)]) for l in (2, 1, 0)])
    assert flat.shape == want.shape and np.abs(flat - want).max() / np.abs(want).max() < 1e-6


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks itself (torch.distributed.run
    children), relays rank 0's JSON line and the exit code.  The kernels are replaced by CPU stand-ins (tests/bench_stub.py:
    gloo, oracle gradients), everything else -- launcher, bucket layout, layer-wise all-reduce, timing protocol,
    max-over-ranks, JSON -- is bench.py's own code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FIL_STUB_OUT"] = str(tmp_path)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--stub", "tests.bench_stub"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0 and res["steps"] == 2
    assert res["config"]["global_batch"] == 24 and res["config"]["parallelism"] == "dp2"
    assert res["rccl"]["world_size_seen"] == 2 and res["rccl"]["backend"] == "gloo" and res["rccl"]["allreduce_bytes"] > 0
    assert "cpu_baseline" not in res          # reported at N = 1 only
    # the reduced bucket: identical on both ranks, equal to the sum of the two shards' gradients
    from tests import bench_stub
    f0, f1 = np.load(tmp_path / "flat0.npy"), np.load(tmp_path / "flat1.npy")
    assert np.array_equal(f0, f1)
    sh = bench_stub.SHAPE
    par = synth.cin_case(sh["batch"], sh["fields"], sh["embed"], sh["conv"], seed=synth.SEED)
    want = None
    for rank in range(2):
        d = synth.cin_case(sh["batch"], sh["fields"], sh["embed"], sh["conv"], seed=synth.SEED + 1 + rank)
        _, dWs, dbs, ddw, ddb = closed.cin_bwd(d["x"], par["Ws"], par["bs"], par["dense_w"], d["g"][:, :1])
        v = np.concatenate([ddw.reshape(-1), ddb.reshape(-1)] + [np.concatenate([dWs[l].reshape(-1), dbs[l].reshape(-1)]) for l in (1, 0)])
        want = v if want is None else want + v
    assert np.abs(f0 - want).max() / np.abs(want).max() < 1e-5
    # a rank count that does not match an existing launcher environment is an error, not a silent single-GPU run
    env_bad = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub", "tests.bench_stub"], env=env_bad,
                       cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_bench_strong_scaling_with_an_uneven_shard(tmp_path):
    """`bench.py --gpus 2 --scaling strong` on a global batch of 13: the ranks take 7 and 6 samples (dp.shard_bounds), the line reports
    the GLOBAL batch once (not 2 x 13), per-rank times for both ranks, the exposed all-reduce time, and the reduced bucket is the sum of
    the two uneven shards' gradients on both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FIL_STUB_OUT=str(tmp_path), FIL_STUB_BATCH="13", PYTHONPATH=root + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong",
                        "--stub", "tests.bench_stub"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["config"]["global_batch"] == 13
    assert res["value"] == pytest.approx(13 * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"]), rel=1e-6)
    rc = res["rccl"]
    # five windows of the same K steps: the first IS the headline, the others only give the spread
    w = res["ms_per_step_windows"]
    assert len(w) == 5 and w[0] == res["ms_per_step"] and res["ms_per_step_min"] <= res["ms_per_step_median"] <= res["ms_per_step_max"]
    assert rc["world_size_seen"] == 2 and len(rc["ms_per_step_per_rank"]) == 2
    assert rc["ms_per_step_min_rank"] <= rc["ms_per_step_max_rank"] and "exposed_allreduce_ms" in rc and rc["allreduce_alone_ms"] > 0
    from tests import bench_stub
    f0, f1 = np.load(tmp_path / "flat0.npy"), np.load(tmp_path / "flat1.npy")
    assert np.array_equal(f0, f1)
    sh = bench_stub.SHAPE
    par = synth.cin_case(13, sh["fields"], sh["embed"], sh["conv"], seed=synth.SEED)
    want = None
    for rank, nb in enumerate((7, 6)):
        d = synth.cin_case(nb, sh["fields"], sh["embed"], sh["conv"], seed=synth.SEED + 1 + rank)
        _, dWs, dbs, ddw, ddb = closed.cin_bwd(d["x"], par["Ws"], par["bs"], par["dense_w"], d["g"][:, :1])
        v = np.concatenate([ddw.reshape(-1), ddb.reshape(-1)] + [np.concatenate([dWs[l].reshape(-1), dbs[l].reshape(-1)]) for l in (1, 0)])
        want = v if want is None else want + v
    assert np.abs(f0 - want).max() / np.abs(want).max() < 1e-5


def _sparse_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, K, B = 50, 4, 12
        rng = np.random.default_rng(5)
        idx = torch.tensor(rng.integers(0, V, B))
        g = torch.tensor(rng.normal(size=(B, K)), dtype=torch.float32)
        lo, hi = dp.shard_bounds(B, rank, world)
        grad = torch.zeros(V, K)
        grad.index_add_(0, idx[lo:hi], g[lo:hi])          # what fil_embed_scatter_add produces for this shard
        rows = dp.exchange_sparse_rows(grad, idx[lo:hi])
        np.save(os.path.join(out_dir, "g%d.npy" % rank), grad.numpy())
        np.save(os.path.join(out_dir, "r%d.npy" % rank), rows.numpy())
    finally:
        dist.destroy_process_group()


def test_sparse_row_exchange_equals_full_batch(tmp_path):
    """Embedding-table gradients: exchanging only the touched rows gives the full-batch gradient, bit-identical on
    every replica (fixed rank order of the additions)."""
    world = 2
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    V, K, B = 50, 4, 12
    rng = np.random.default_rng(5)
    idx = torch.tensor(rng.integers(0, V, B))
    g = torch.tensor(rng.normal(size=(B, K)), dtype=torch.float32)
    want = torch.zeros(V, K)
    want.index_add_(0, idx, g)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)
    assert np.abs(g0 - want.numpy()).max() < 1e-6
    assert np.array_equal(np.load(tmp_path / "r0.npy"), np.unique(idx.numpy()))


def test_sparse_row_exchange_single_process_is_a_noop():
    grad = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    keep = grad.clone()
    rows = dp.exchange_sparse_rows(grad, torch.tensor([4, 1, 4]))
    assert torch.equal(grad, keep) and rows.tolist() == [1, 4]


def _sparse_reg_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, K, B, reg = 50, 4, 12, 0.25
        rng = np.random.default_rng(6)
        table = torch.tensor(rng.normal(size=(V, K)), dtype=torch.float32)
        idx = torch.tensor(rng.integers(0, V // 2, B))        # rows >= V/2 are touched by nobody
        g = torch.tensor(rng.normal(size=(B, K)), dtype=torch.float32)
        lo, hi = dp.shard_bounds(B, rank, world)
        grad = torch.zeros(V, K)
        grad.index_add_(0, idx[lo:hi], g[lo:hi] / world)      # the shard's part of the mean loss
        dp.exchange_sparse_rows(grad, idx[lo:hi])
        dp.add_table_l2_grad_(grad, table, [(0, 30, reg), (40, 50, 2 * reg)])
        np.save(os.path.join(out_dir, "g%d.npy" % rank), grad.numpy())
    finally:
        dist.destroy_process_group()


def test_sparse_row_exchange_with_table_regulariser(tmp_path):
    """emb_reg > 0: the l2 term of a table is added after the exchange, once, identically on every replica --
    replicas stay bit-identical and rows nobody touched get exactly 2*reg*w (not reg/world)."""
    world = 2
    mp.spawn(_sparse_reg_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    V, K, B, reg = 50, 4, 12, 0.25
    rng = np.random.default_rng(6)
    table = torch.tensor(rng.normal(size=(V, K)), dtype=torch.float32, requires_grad=True)
    idx = torch.tensor(rng.integers(0, V // 2, B))
    g = torch.tensor(rng.normal(size=(B, K)), dtype=torch.float32)
    loss = (table[idx] * g).sum() / world + reg * table[0:30].square().sum() + 2 * reg * table[40:50].square().sum()
    loss.backward()
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1)
    assert np.abs(g0 - table.grad.numpy()).max() < 1e-6
    assert np.abs(g0[45] - 4 * reg * table.detach().numpy()[45]).max() < 1e-7 and not g0[30:40].any()
