"""Data parallelism for the interaction layers: one process per GPU, batch rows sharded, ONE collective per step.

Every layer on the hot path is independent across the batch axis, so rank r takes rows
[r*B/N, (r+1)*B/N) of the inputs and of the upstream gradient, parameters are replicated, and the only exchange is
a sum all-reduce of the dense parameter gradients (RCCL over xGMI through torch.distributed's "nccl" backend;
"gloo" in the CPU tests).  Gradients are written by the kernels straight into one flat fp32 bucket, so the
all-reduce is a single call with no packing copies (CIN at the north-star shape: 1,473,073 floats = 5.9 MB).
The reference has no distributed code at all; this module is net-new.
"""
import torch
import torch.distributed as dist

from .streams import independent_stream


def shard_bounds(n_rows, rank, world):
    """Contiguous, balanced row range of `rank`: sizes differ by at most one row."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(t, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


class GradBucket:
    """One flat fp32 buffer holding the gradients of `params` back to back; .views[i] aliases params[i]'s slot."""

    def __init__(self, shapes, device):
        self.shapes = [tuple(s) for s in shapes]
        sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        self.views = []
        off = 0
        for s, n in zip(self.shapes, sizes):
            self.views.append(self.flat[off:off + n].view(s))
            off += n

    @classmethod
    def for_params(cls, params):
        params = list(params)
        return cls([p.shape for p in params], params[0].device)

    def nbytes(self):
        return self.flat.numel() * 4

    def all_reduce(self, group=None, async_op=False):
        """Sum over the data-parallel ranks (gradients of a summed loss add up; divide by world for a mean loss)."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    def copy_from_grads(self, params):
        for v, p in zip(self.views, params):
            v.copy_(p.grad if p.grad is not None else torch.zeros_like(v))

    def assign_to_grads(self, params):
        for v, p in zip(self.views, params):
            p.grad = v


class LayerwiseAllReduce:
    """All-reduce of a flat gradient bucket in SEGMENTS, each started as soon as its gradients are final.

    The CIN backward produces parameter gradients from the top layer down, each before the (long) data-gradient kernel
    of its layer (fil.h: grad_ready_events).  `segments` are element ranges [lo, hi) of `flat` in that order;
    `events[i]` is the torch.cuda.Event the library records when segment i is final.  launch() -- called right after the
    backward has been ENQUEUED -- makes a side stream wait on each event and issues that segment's all-reduce there, so
    the collectives (RCCL kernels, a handful of workgroups) run underneath the remaining backward kernels; wait() joins
    them into the current stream.  On CPU tensors (gloo tests) there are no streams: launch() reduces synchronously.
    Sums are over ranks in the backend's fixed order, segment boundaries never split a reduction, so the result is the
    same as one all-reduce of the whole bucket."""

    def __init__(self, flat, segments, group=None, force=False):
        """force=True: issue the collectives even in a one-rank group (an all-reduce over one rank leaves the data unchanged but
        runs the whole path -- side stream, event waits, the backend's kernels: how the overlap machinery is exercised and timed
        on a single GPU, bench.py --force-collective)."""
        self.flat, self.segments, self.group = flat, [(int(a), int(b)) for a, b in segments], group
        self.force = bool(force)
        assert all(0 <= a <= b <= flat.numel() for a, b in self.segments)
        self.cuda = flat.is_cuda
        self.events = [torch.cuda.Event() for _ in self.segments] if self.cuda else [None] * len(self.segments)
        # (a stream that really runs beside the compute stream: one sharing its hardware queue would stall it at every wait)
        self.side = independent_stream(flat.device, tag="grad_allreduce") if self.cuda else None
        self._works = []

    def active(self):
        return dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def launch(self):
        if not self.active():
            return
        if not self.cuda:
            for lo, hi in self.segments:
                if hi > lo:
                    dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            return
        # Every cross-stream dependency is a barrier packet plus a signal round trip on the waiting queue (~15 us each on this
        # stack): the per-collective joins go to the SIDE stream (work.wait() there makes it wait for the backend's internal
        # stream), the compute stream joins the side stream once, in wait().
        with torch.cuda.stream(self.side):
            for ev, (lo, hi) in zip(self.events, self.segments):
                if hi <= lo:
                    continue
                self.side.wait_event(ev)
                w = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                w.wait()
                self._works.append(w)

    def wait(self):
        """Make the current stream wait for every segment's collective (call before anything reads the gradients)."""
        if not self.cuda:
            for w in self._works:
                w.wait()
        self._works = []
        if self.cuda and self.active():
            torch.cuda.current_stream().wait_stream(self.side)


def cin_bucket_layout(W_shapes, b_shapes, head_shapes):
    """Flat gradient bucket of a CIN in the order the backward finishes them: [head | layer L-1 | ... | layer 0], each
    layer as (dW_l, dbias_l).  Returns (sizes, segments, index) where segments[i] is the element range that becomes
    final with event i (segment 0 = head + top layer, then one per layer downwards) and index maps
    ("W", l) / ("b", l) / ("head", j) to its position in `sizes`."""
    L = len(W_shapes)
    sizes, index = [], {}
    for j, sh in enumerate(head_shapes):
        index[("head", j)] = len(sizes)
        sizes.append(int(torch.Size(sh).numel()))
    bounds = []
    for l in range(L - 1, -1, -1):
        index[("W", l)] = len(sizes)
        sizes.append(int(torch.Size(W_shapes[l]).numel()))
        index[("b", l)] = len(sizes)
        sizes.append(int(torch.Size(b_shapes[l]).numel()))
        bounds.append(sum(sizes))
    segments, lo = [], 0
    for hi in bounds:
        segments.append((lo, hi))
        lo = hi
    return sizes, segments, index


def merge_segments_by_point(segments, points):
    """Coalesce the per-layer segments of cin_bucket_layout whose gradients become final at the same point of the backward.

    segments[i] belongs to layer L-1-i (segment 0 also holds the head); points = functional.cin_grad_ready_points(...) (L+1
    ordinals, [l] per layer).  Returns (merged segments, layer_of_event): merged segment j is reduced behind the grad_ready slot
    of layer layer_of_event[j] -- the lowest layer of its group, whose slot is recorded when the whole group is final -- and the
    merged segments come in the order their events fire.  Every
    collective costs tens of microseconds of stream plumbing whatever its size: with the fused tail the two top layers finish
    together and share one."""
    L = len(segments)
    if points is None:      # the ranks do not agree on the points (agree_on_points): bucket order, one collective per layer
        return list(segments), list(range(L - 1, -1, -1))
    merged, layer_of_event = [], []
    for i, (lo, hi) in enumerate(segments):
        l = L - 1 - i
        if merged and points[l] == points[layer_of_event[-1]]:
            merged[-1] = (merged[-1][0], hi)
            layer_of_event[-1] = l
        else:
            merged.append((lo, hi))
            layer_of_event.append(l)
    # issue order = readiness order: the merged quadratic tail finishes the FIRST layer's gradients before the top two layers'
    # (points [0, 1, 1, 0]: the head's come out with layer 0's); a collective queued behind a later event would wait for it although its own data is final
    order = sorted(range(len(merged)), key=lambda i: points[layer_of_event[i]])
    return [merged[i] for i in order], [layer_of_event[i] for i in order]


def agree_on_points(points, device=None, group=None):
    """The readiness points decide which segments merge and in which order the collectives are issued, and they depend on each
    rank's LOCAL batch (the library switches tails at 16,384 rows: fil_cin_grad_ready_points gives [0,1,1,0] above and [2,1,1,0] at
    or below it), so with uneven strong-scaling shards that straddle the threshold (global 4097 rows over 4 ranks: 1025 / 1024 /
    1024 / 1024) the ranks would issue collectives of different sizes in different orders -- an RCCL hang or silently wrong sums.
    Every rank contributes its vector; the result is `points` when ALL ranks hold the same one and None otherwise, which
    merge_segments_by_point answers with the rank-invariant fallback (per-layer segments in bucket order, each behind its own
    layer's event: correct on every rank whatever its tail, a little less overlap).  Collective: call it on every rank."""
    points = [int(p) for p in points]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return points
    mine = torch.tensor(points, dtype=torch.int64, device=device if device is not None else "cpu")
    every = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(every, mine, group=group)
    return points if all(bool(torch.equal(e, mine)) for e in every) else None


def allreduce_module_grads(module, group=None):
    """Bucketed all-reduce of the gradients of every parameter of `module` (after backward)."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return None
    bucket = GradBucket.for_params(params)
    bucket.copy_from_grads(params)
    bucket.all_reduce(group)
    bucket.assign_to_grads(params)
    return bucket


def exchange_sparse_rows(grad, touched_rows, group=None):
    """Data-parallel sum of an embedding-table gradient that is zero outside the rows each rank touched.

    grad [V, K] holds this rank's scatter-added gradient (rows outside `touched_rows` are zero); `touched_rows` is a
    1-D int64 tensor of the table rows this rank's batch shard indexed (duplicates allowed).  Instead of all-reducing
    the whole table (V*K floats: 100s of MB for Criteo-size vocabularies), every rank contributes only its unique
    touched rows: counts are all-gathered, then the padded (row id, row value) lists, and each rank adds the lists in
    rank order -- the same fixed order everywhere, so all replicas end with bit-identical gradients.
    Returns the sorted union of touched rows (the only rows an optimizer needs to visit).  No-op for world size 1.
    """
    rows = torch.unique(touched_rows.reshape(-1).to(torch.int64))
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    n_local = torch.tensor([rows.numel()], dtype=torch.int64, device=grad.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    k = grad.shape[1]
    pad_rows = torch.zeros(n_max, dtype=torch.int64, device=grad.device)
    pad_vals = torch.zeros((n_max, k), dtype=grad.dtype, device=grad.device)
    pad_rows[:rows.numel()] = rows
    pad_vals[:rows.numel()] = grad[rows]
    all_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    all_vals = [torch.empty_like(pad_vals) for _ in range(world)]
    dist.all_gather(all_rows, pad_rows, group=group)
    dist.all_gather(all_vals, pad_vals, group=group)
    grad[rows] = 0
    for r in range(world):  # rank order: identical summation order on every replica
        grad.index_add_(0, all_rows[r][:counts[r]], all_vals[r][:counts[r]])
    return torch.unique(torch.cat([all_rows[r][:counts[r]] for r in range(world)]))


def add_table_l2_grad_(grad, table, ranges):
    """grad[lo:hi] += 2 * reg * table[lo:hi] for every (lo, hi, reg): the gradient of Keras' l2(reg) = reg * sum(w^2) on those rows
    (SparseEmbed.table_l2_ranges()).  Called by the data-parallel trainer AFTER exchange_sparse_rows, with the same table on
    every replica: the term is then added exactly once, identically everywhere, and the exchanged gradient stayed sparse.
    (Through autograd on a loss divided by the world size it would reach a row touched by one rank as reg/w there but 2*reg/w
    on a rank that did not touch it: replicas drift apart.)"""
    with torch.no_grad():
        for lo, hi, reg in ranges:
            grad[lo:hi].add_(table[lo:hi], alpha=2.0 * float(reg))
    return grad
