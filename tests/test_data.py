"""ml_function_amd.data: the reference's tf.data input pipeline (data_prepare.py:335-337) and static_batch (:393-406)
re-stated; CPU only."""
import time

import numpy as np
import pytest

from ml_function_amd import data


def test_shuffle_buffer_semantics():
    rng = np.random.default_rng(0)
    n, buf = 10000, 2048
    order = data.shuffled_indices(n, buf, rng)
    assert sorted(order.tolist()) == list(range(n))                   # a permutation
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    assert (pos >= np.arange(n) - (buf - 1)).all()                    # element i cannot leave before output i - buffer + 1
    assert (pos[:buf] < buf * 12).mean() > 0.95 and pos.std() > 100   # really shuffled, but only locally
    assert np.array_equal(data.shuffled_indices(7, 1, rng), np.arange(7))          # buffer 1 = no shuffle
    assert sorted(data.shuffled_indices(5, 2048, rng).tolist()) == [0, 1, 2, 3, 4]  # buffer larger than the data


def test_pipeline_repeat_then_batch():
    n, bs = 1000, 64
    x = np.arange(n, dtype=np.int64)
    ds = ({"dense": x.astype(np.float32)[:, None], "ids": np.stack([x, x + 1], 1)}, x % 2)
    pipe = data.data_pipeline(ds, batch_size=bs, seed=1)
    batches = list(pipe)
    assert len(batches) == len(pipe) == -(-2 * n // bs)
    assert all(len(b[1]) == bs for b in batches[:-1]) and len(batches[-1][1]) == 2 * n - bs * (len(batches) - 1)
    seen = np.concatenate([b[0]["ids"][:, 0] for b in batches])
    assert np.array_equal(np.bincount(seen, minlength=n), np.full(n, 2))             # every row exactly `repeat` times
    first, second = seen[:n], seen[n:]
    assert sorted(first.tolist()) == list(range(n)) and not np.array_equal(first, second)   # reshuffled each repetition
    b = batches[3]
    assert np.array_equal(b[0]["dense"][:, 0], b[0]["ids"][:, 0]) and np.array_equal(b[1], b[0]["ids"][:, 0] % 2)   # rows stay aligned
    assert np.array_equal(np.concatenate([bb[1] for bb in data.data_pipeline(ds, bs, seed=1)]),
                          np.concatenate([bb[1] for bb in batches]))                 # seeded: reproducible
    with pytest.raises(ValueError):
        data.data_pipeline((x, x[:5]), 4)


def test_prefetch_runs_ahead_of_the_consumer():
    x = np.arange(64)
    pipe = data.data_pipeline((x,), batch_size=8, shuffle_buffer=0, repeat=1, prefetch=2, device="cpu")
    it = iter(pipe)
    first = next(it)
    time.sleep(0.2)                                   # the producer fills its queue while the consumer is busy
    assert int(first[0][0]) == 0 and first[0].dtype.is_floating_point is False
    rest = [b[0] for b in it]
    assert np.array_equal(np.concatenate([first[0].numpy()] + [r.numpy() for r in rest]), x)


def test_static_batch_is_a_bootstrap_resample():
    rng = np.random.default_rng(4)
    df = {"a": np.arange(1000), "b": np.arange(1000) * 2}
    out = data.static_batch(df, batch_size=64, use_shuffle=True, rng=rng)
    assert len(out["a"]) == (1000 // 64) * 64 and np.array_equal(out["b"], out["a"] * 2)
    assert len(np.unique(out["a"])) < len(out["a"])                  # with replacement, like np.random.choice's default
    arr = data.static_batch(list(range(100)), batch_size=30, use_shuffle=False, rng=rng)
    assert arr.shape == (90,)


@pytest.mark.gpu
def test_prefetched_batches_survive_queued_consumer_work():
    """Batches are allocated and filled on the producer's side stream.  The consumer queues a long kernel, then reads the batch
    on its own stream and drops it at once (the training loop's pattern): without record_stream on the consumer's stream the
    caching allocator may hand the block back to the producer, whose next copy overwrites data the queued reads still need."""
    import torch
    dev = torch.device("cuda", 0)
    n, bs = 64 * 4096, 4096
    x = np.arange(n, dtype=np.int64)
    pipe = data.data_pipeline((x, (x * 3).astype(np.float32)), batch_size=bs, shuffle_buffer=0, repeat=1, prefetch=2, device=dev)
    busy = torch.randn(4096, 4096, device=dev)
    sums, fsums = [], []
    for ids, vals in pipe:
        for _ in range(6):                      # ~ms of queued work in front of every read of the batch
            busy = torch.tanh(busy @ busy * 1e-3)
        sums.append(ids.sum())
        fsums.append((vals.double() - 3.0 * ids.double()).abs().max())
        del ids, vals                           # block returns to the allocator while the reads are still queued
    torch.cuda.synchronize()
    got = torch.stack(sums).cpu().numpy()
    want = x.reshape(-1, bs).sum(1)
    assert np.array_equal(got, want)
    assert float(torch.stack(fsums).max()) == 0.0
