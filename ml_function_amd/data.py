"""Batching with the semantics of the reference's input pipeline (kon/utils/data_prepare.py):

  data_pipeline (:335-337)   tf.data.Dataset.from_tensor_slices(dataSet).shuffle(2048).repeat(2).batch(batch_size).prefetch(2)
  static_batch  (:393-406)   a bootstrap resample of (n // batch_size) * batch_size rows (np.random.choice WITH replacement,
                             the reference's default), optionally shuffled

re-stated without TensorFlow.  What is kept is the documented behaviour of each tf.data stage, not TF's random stream:
  * from_tensor_slices: any nesting of tuples / lists / dicts of arrays, sliced along axis 0;
  * shuffle(buffer): the buffer is filled with the first `buffer` elements; every output picks one buffered element
    uniformly at random and refills its slot with the next input element (so element i can leave at output position
    i - buffer + 1 at the earliest); reshuffled on every repetition (reshuffle_each_iteration, TF's default);
  * repeat(n) BEFORE batch: batches run across the epoch boundary, only the very last batch can be short;
  * batch(batch_size): drop_remainder=False;
  * prefetch(depth): a producer thread keeps `depth` finished batches ahead of the consumer; on a GPU device the batches
    are copied host -> device from pinned memory on a side stream, and the consumer's stream waits on the copy's event.
"""
import queue
import threading

import numpy as np
import torch

from .streams import independent_stream


def _map_structure(fn, s):
    if isinstance(s, dict):
        return {k: _map_structure(fn, v) for k, v in s.items()}
    if isinstance(s, (tuple, list)):
        return type(s)(_map_structure(fn, v) for v in s)
    return fn(s)


def _leaves(s):
    if isinstance(s, dict):
        for v in s.values():
            yield from _leaves(v)
    elif isinstance(s, (tuple, list)):
        for v in s:
            yield from _leaves(v)
    else:
        yield s


def shuffled_indices(n, buffer_size, rng):
    """Order in which tf.data's shuffle(buffer_size) emits n elements (see module docstring), as an int64 array."""
    out = np.empty(n, dtype=np.int64)
    buf = list(range(min(buffer_size, n)))
    nxt = len(buf)
    for pos in range(n):
        j = int(rng.integers(0, len(buf)))
        out[pos] = buf[j]
        if nxt < n:
            buf[j] = nxt
            nxt += 1
        else:
            buf[j] = buf[-1]
            buf.pop()
    return out


def index_stream(n, buffer_size=2048, repeat=2, seed=None):
    """from_tensor_slices(...).shuffle(buffer_size).repeat(repeat): the element indices in output order."""
    rng = np.random.default_rng(seed)
    for _ in range(repeat):
        yield from shuffled_indices(n, buffer_size, rng) if buffer_size and buffer_size > 1 else np.arange(n, dtype=np.int64)


class data_pipeline:
    """Iterable over batches: shuffle(shuffle_buffer).repeat(repeat).batch(batch_size).prefetch(prefetch) of `dataset`
    (a nested structure of equally long arrays).  device: batches are torch tensors there (None: numpy arrays)."""

    def __init__(self, dataset, batch_size, shuffle_buffer=2048, repeat=2, prefetch=2, seed=None, device=None):
        self.dataset = _map_structure(np.asarray, dataset)
        sizes = {len(a) for a in _leaves(self.dataset)}
        if len(sizes) != 1:
            raise ValueError("from_tensor_slices: all components must have the same first dimension, got %s" % sorted(sizes))
        self.n = sizes.pop()
        self.batch_size, self.shuffle_buffer, self.repeat, self.prefetch = int(batch_size), shuffle_buffer, int(repeat), int(prefetch)
        self.seed, self.device = seed, None if device is None else torch.device(device)

    def __len__(self):
        return -(-self.n * self.repeat // self.batch_size)

    def _batches(self):
        idx = np.fromiter(index_stream(self.n, self.shuffle_buffer, self.repeat, self.seed), dtype=np.int64, count=self.n * self.repeat)
        for lo in range(0, len(idx), self.batch_size):
            sel = idx[lo:lo + self.batch_size]
            yield _map_structure(lambda a: a[sel], self.dataset)

    def _to_device(self, batch, stream):
        if self.device is None:
            return batch, None
        if self.device.type != "cuda":
            return _map_structure(lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device), batch), None
        with torch.cuda.stream(stream):
            out = _map_structure(lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(self.device, non_blocking=True), batch)
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev

    def __iter__(self):
        if self.prefetch <= 0:
            for b in self._batches():
                out, _ = self._to_device(b, None if self.device is None or self.device.type != "cuda" else torch.cuda.current_stream())
                yield out
            return
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()
        # (the copy stream must not share a hardware queue with the compute stream, or the copies queue up behind its kernels)
        stream = independent_stream(self.device, tag="data_pipeline") if self.device is not None and self.device.type == "cuda" else None

        def produce():
            try:
                for b in self._batches():
                    item = self._to_device(b, stream)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:   # surfaced in the consumer
                q.put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                out, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream()
                    cur.wait_event(ev)
                    # The batch was allocated on the producer's side stream: tell the caching allocator that the consumer's
                    # stream uses it too, otherwise a dropped batch's block goes back to the side-stream pool at once and the
                    # producer's next non_blocking copy may overwrite memory that queued consumer kernels still read.
                    for t in _leaves(out):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(cur)
                yield out
        finally:
            stop.set()


def static_batch(df, batch_size, use_shuffle=True, rng=None):
    """DataGenerater.static_batch (data_prepare.py:393-406): (n // batch_size) * batch_size row indices drawn by
    np.random.choice(range(n), size=...) -- i.e. WITH replacement, a bootstrap resample, the reference's behaviour -- shuffled
    when use_shuffle, applied to an array / list or to every value of a dict."""
    rng = np.random.default_rng() if rng is None else rng
    n = np.asarray(df[list(df.keys())[0]]).shape[0] if isinstance(df, dict) else len(df)
    batch_num = (n // batch_size) * batch_size
    need_idx = rng.choice(n, size=batch_num)
    if use_shuffle:
        rng.shuffle(need_idx)
    if isinstance(df, dict):
        return {k: np.asarray(v)[need_idx] for k, v in df.items()}
    return np.asarray(df)[need_idx]
