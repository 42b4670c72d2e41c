#!/usr/bin/env python
"""Headline benchmark: samples/s of xDeepFM-CIN forward+backward (BASELINE.json configs[3]).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  For N > 1 either launch it with `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`, or just run `python bench.py --gpus N`: without WORLD_SIZE in the environment the parent starts the N ranks
itself (fresh child processes through torch.distributed.run, before anything in the parent touches a GPU) and relays
rank 0's JSON line and the exit code.
A step = CIN forward + backward over one synthetic batch of B=4096 samples per GPU (F=39, K=16, 3x128 feature
maps, fp32, all parameter and input gradients) + (N > 1) the RCCL all-reduce of the parameter gradients, issued per
layer on a side stream as each layer's gradients become final (dp.LayerwiseAllReduce), joined before the step ends.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task statement),
including `roofline` for the dominant kernel (HIP-event timing on the launch stream via fil_profile_begin/_end)
and `cpu_baseline` (the oracle's op-for-op torch-CPU restatement of the reference TF2 graph, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, F, K = 4096, 39, 16
CONV = [128, 128, 128]
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md dense fp16/bf16 MFMA peak (16x the fp32 matrix rate)


def make_inputs(rank, device, batch=B_PER_GPU, conv=CONV, fields=F, embed=K, param_batch=None):
    from ml_function_amd import synth
    # parameters: identical on every rank -- drawn for `param_batch` samples (the generator draws the inputs first, so the weights
    # depend on the batch size: with uneven strong-scaling shards every rank must ask for the same one, the global batch)
    c = synth.cin_case(param_batch or batch, fields, embed, conv, seed=synth.SEED)
    d = synth.cin_case(batch, fields, embed, conv, seed=synth.SEED + 1 + rank)  # data shard: per rank
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    return dict(x=t(d["x"]), g=t(d["g"][:, 0]), Ws=[t(w) for w in c["Ws"]], bs=[t(b) for b in c["bs"]],
                dense_w=t(c["dense_w"]), dense_b=t(c["dense_b"]))


def make_bucket(inp, device):
    """One flat fp32 gradient bucket laid out in the order the backward finishes the gradients ([head | top layer | ...
    | layer 1], dp.cin_bucket_layout); dW/db/ddense are views into it (kernels write in place, no packing copies).
    Returns (flat, grads dict for Fn.cin_backward_raw, segments for dp.LayerwiseAllReduce)."""
    from ml_function_amd import dp
    L = len(inp["Ws"])
    sizes, segments, index = dp.cin_bucket_layout([w.shape for w in inp["Ws"]], [b.shape for b in inp["bs"]],
                                                  [inp["dense_w"].shape, (1,)])
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    view = lambda key: flat[offs[index[key]]:offs[index[key] + 1]]
    grads = dict(dx=torch.empty_like(inp["x"]),
                 dW=[view(("W", l)).view_as(inp["Ws"][l]) for l in range(L)],
                 db=[view(("b", l)) for l in range(L)], ddw=view(("head", 0)).view(-1, 1), ddb=view(("head", 1)))
    return flat, grads, segments


def usable_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(sample_b=B_PER_GPU, steps=3):
    """Reference-graph restatement (oracle/graph.py) on the host CPU, fp32, all usable cores: the benchmark's own batch
    (B=4096, BASELINE.md section 2 protocol), 1 warm-up + 3 timed steps (a step is seconds of CPU work), median."""
    from ml_function_amd import synth
    from oracle import graph
    torch.set_num_threads(usable_cpus())
    c = synth.cin_case(sample_b, F, K, CONV)
    mk = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
    x, Ws, bs = mk(c["x"]), [mk(w) for w in c["Ws"]], [mk(b) for b in c["bs"]]
    dw, db = mk(c["dense_w"]), mk(c["dense_b"])
    g = torch.tensor(c["g"])
    times = []
    for i in range(steps + 1):
        for p in [x, dw, db] + Ws + bs:
            p.grad = None
        t0 = time.perf_counter()
        out = graph.cin(x, Ws, bs, dw, db, output_dim=1)
        out.backward(g)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    return dict(value=sample_b / med, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample="the full batch B=%d of the same config (F=39,K=16,3x128), %d timed fwd+bwd steps after 1 warm-up (median %.2f s "
                       "per step), fp32, op-for-op torch-CPU restatement of the reference TF2 graph with the outer product "
                       "materialised as TF would (TF not installable)" % (sample_b, steps, med))


PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured with a float4 copy)


def deepfm_benchmark(args):
    """Whole-model steps.  deepfm = BASELINE config 2: DeepFM (FM + 2-layer MLP), 39 fields, K=16, B=4096, bf16 (embedding
    gather, FM layer on bf16 embeddings, MLP under bf16 autocast, softmax head, BCE, backward incl. the embedding
    scatter-add).  xdeepfm = the north-star layer inside its model (models.XDeepFM: linear + CIN 3x128 + 256-128-64 MLP,
    fp32).  Eager and replayed from a HIP graph."""
    from ml_function_amd import losses, models
    dev = torch.device("cuda", 0)
    B = args.batch or 4096
    rng = np.random.default_rng(2020)
    vocab = [int(v) for v in np.exp(rng.uniform(np.log(10), np.log(1e5), 39))]
    xd = args.workload == "xdeepfm"
    fi = models.FeatureInput(sparseInfo=models.make_sparse_info(vocab, embed_dim=16), useLinear=True, useAddLinear=xd,
                             useFlattenLinear=xd, emitXT=xd,   # (xDeepFM: the gather also emits the layout the CIN kernels read)
                             embedDtype=None if xd else torch.bfloat16)   # (DeepFM c2: the embedding block in bf16 straight out of the gather)
    body = models.XDeepFM(conv_size=[128, 128, 128]) if xd else models.DeepFM(hidden_units=[256, 128])

    class Bf16Body(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, fea):
            from ml_function_amd.layers.base import merge_packed_views
            if fea.sparse_embed[0].dtype != torch.bfloat16:   # (a gather that emits fp32: one cast of the packed block, no copies)
                blk = merge_packed_views(list(fea.sparse_embed))
                fea.sparse_embed = (list(blk[0].bfloat16().split(1, dim=1)) if len(blk) == 1
                                    else [e.bfloat16() for e in fea.sparse_embed])
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                return self.inner(fea).float()

    torch.manual_seed(0)
    model = models.CTRModel(fi, body if xd else Bf16Body(body)).to(dev)
    dense = torch.tensor(rng.random((B, 13), dtype=np.float32), device=dev)
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in vocab], 1), device=dev)
    y = torch.tensor(rng.integers(0, 2, B), dtype=torch.float32, device=dev)
    model(dense, idx)  # lazy weight creation
    params = [p for p in model.parameters()]

    def step():
        for p in params:
            p.grad = None
        out = model(dense, idx)
        losses.binary_crossentropy(out[:, -1], y, eps=1e-6).backward()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graph_ms = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for _ in range(args.warmup):
            gr.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            gr.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t1) / args.steps * 1e3
    name = ("xDeepFM whole model (linear + CIN 3x128 + MLP 256-128-64) fwd+bwd, 39 fields K=16, fp32, B=%d" % B if xd else
            "DeepFM (FM + MLP 256-128) whole-model fwd+bwd, 39 fields K=16, bf16, B=%d (BASELINE.json configs[1])" % B)
    return {"metric": "samples/sec fwd+bwd " + name, "value": B * args.steps / dt, "unit": "samples/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if xd else "bf16 (fp32 accumulate)", "data": "synthetic",
            "config": {"workload": name}, "hipgraph_replay_ms_per_step": graph_ms,
            "hipgraph_samples_per_s": (B / (graph_ms * 1e-3)) if graph_ms else None}


def side_benchmark(args):
    """FM (config 2), DCN (config 3), AutoInt interacting layer (config 5): one GPU, fwd+bwd, same timing protocol."""
    from ml_function_amd import _lib, synth
    from ml_function_amd import functional as Fn
    dev = torch.device("cuda", 0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    if args.workload == "fm":
        B = args.batch or 4096
        c = synth.fm_case(B, 39, 16)
        emb, lin, g = t(c["emb"]).requires_grad_(), t(c["lin"]).requires_grad_(), t(c["g"])
        def step():
            emb.grad = lin.grad = None
            Fn.fm(emb, lin).backward(g)
        name, kernels, bound = "FM 2nd-order fwd+bwd F=39 K=16 fp32 B=%d" % B, ("fm_fwd", "fm_bwd"), "hbm"
    elif args.workload == "dcn":
        B = args.batch or 8192
        c = synth.dcn_case(B, 1248, 3)
        x, w, b, g = t(c["x"]).requires_grad_(), t(c["w"]).requires_grad_(), t(c["b"]).requires_grad_(), t(c["g"])
        def step():
            x.grad = w.grad = b.grad = None
            Fn.dcn_cross(x, w, b).backward(g)
        name, kernels, bound = "DCN 3 cross layers fwd+bwd D=1248 fp32 B=%d" % B, ("dcn_fwd", "dcn_bwd"), "hbm"
    else:
        B = args.batch or 4096
        Fa, Ka, Ha, Aa, L = 200, 16, 4, 16, args.layers
        c = synth.attn_stack_case(B, Fa, Ka, Ha, Aa, L)
        xin = t(c["x"]).requires_grad_()
        layers = [tuple(t(p).requires_grad_() for p in lay) for lay in c["layers"]]
        dy = t(c["dy"])
        def step():
            xin.grad = None
            for lay in layers:
                for p in lay:
                    p.grad = None
            Fn.autoint_stack(xin, layers, precision=args.precision).backward(dy)
        name = "AutoInt %d interacting layer%s fwd+bwd F=200 K=16 H=4 A=16 %s B=%d" % (L, "s" if L > 1 else "", args.precision, B)
        bound, kernels = "hbm", ("attn_fwd", "attn_bwd")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # the timed loop carries no profiler events (an event pair is 5-10 us of stream time per kernel: for these two-kernel steps that
    # was a third of the "eager" time); the per-kernel table comes from a second, untimed pass
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_begin()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    graph_ms = None
    if args.graph:
        # the launch-bound shapes (FM / DCN at their BASELINE batch sizes): the same step captured once into a HIP graph
        # (torch.cuda.CUDAGraph; the C ABI neither allocates nor synchronises) and replayed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        for _ in range(args.warmup):
            gr.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            gr.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t1) / args.steps * 1e3
    ks = {k: v for k, v in prof.items() if k in kernels}
    dom = max(ks, key=lambda k: ks[k]["total_ms"])
    d = ks[dom]
    f16 = args.workload == "autoint" and args.precision == "f16_mfma"
    extra = {}
    if args.workload == "autoint":
        # What binds these kernels: HBM passes over the [H,B,F,A] tensors and the sigmoid issue rate,
        # not the matrix pipe.  Algorithmic bytes per launch (fp32): forward = x in, y + saved av out; backward = x, av, y,
        # dy in, dx out.  Averaged over the L layers of the stack (layers 2.. read K = H*A = 64 input features).
        hbfa = Ha * B * Fa * Aa * 4.0
        xin_b = [B * Fa * (Ka if l == 0 else Ha * Aa) * 4.0 for l in range(L)]
        bytes_fwd = sum(xb + 2 * hbfa for xb in xin_b) / L
        bytes_bwd = sum(2 * xb + 3 * hbfa for xb in xin_b) / L
        nbytes = bytes_bwd if dom == "attn_bwd" else bytes_fwd
        rate = nbytes / (d["avg_ms"] * 1e-3)
        # vector-issue roofline of the score pass, per 16x16 tile of one wave, priced with the issue costs of
        # MI355X_MICROARCH.md (v_exp_f32 / v_rcp_f32 8 cycles, plain VALU 4, v_cvt_pk 4-5, an MFMA holds the vector issue 8):
        #   fwd  sigmoid = 4 exp + 4 rcp + 4 add; 2 cvt_pk (S' as the fp16 operand); 2 MFMA (S', av)
        #   bwd  the same sigmoid; 12 mul (dS S (1-S)); 4 cvt_pk (S, dP); 5 MFMA (S, dS, dq, 2 x dk)
        # (LayerNorm, projections, conversions of the block prologue are NOT counted: this is the floor of the tile loop alone)
        cyc_tile = {"attn_fwd": 8 * 8 + 4 * 4 + 2 * 4.5 + 2 * 8, "attn_bwd": 8 * 8 + (4 + 12) * 4 + 4 * 4.5 + 5 * 8}
        tiles = Ha * B * (Fa / 16.0) ** 2
        floor_ms = {k: tiles * c / (1024 * 2.4e9) * 1e3 for k, c in cyc_tile.items()}
        extra = {"valu_roofline": {"what": "vector-issue cycles of one pass over the F x F scores (per 16x16 tile: transcendentals x 8 + "
                                           "other VALU x 4 + cvt_pk x 4.5 + MFMA issue x 8; 1024 SIMDs x 2.4 GHz)",
                                   "kernel": dom, "cycles_per_tile": cyc_tile, "tiles_per_launch": tiles,
                                   "floor_ms": floor_ms, "measured_ms": {k: v["avg_ms"] for k, v in ks.items()},
                                   "frac": floor_ms[dom] / d["avg_ms"],
                                   "frac_per_kernel": {k: floor_ms[k] / v["avg_ms"] for k, v in ks.items() if k in floor_ms},
                                   # the same mixes MEASURED on the chip (tools/probe_tile.hip, profiles/r05_probe_tile.txt: no memory, no
                                   # LDS, 2-4 waves per SIMD -- occupancy does not change them): a transcendental is 10.7-12.4 cycles per
                                   # wave instruction and SIMD, not 8, and the MFMAs' issue does not hide; f16 mode only
                                   "probe_cycles_per_tile": {"attn_fwd": 115.0, "attn_bwd": 200.0},
                                   "frac_of_probe_floor": {k: tiles * c / (1024 * 2.4e9) * 1e3 / ks[k]["avg_ms"]
                                                           for k, c in (("attn_fwd", 115.0), ("attn_bwd", 200.0)) if k in ks} if f16 else None},
                 "mfma_tflops": {k: v["work"] / (v["avg_ms"] * 1e-3) / 1e12 for k, v in ks.items()},
                 "mfma_frac_of_peak": {k: v["work"] / (v["avg_ms"] * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS)
                                       for k, v in ks.items()},
                 "ms_per_layer": dt / args.steps * 1e3 / L}
    else:
        rate = d["work"] / (d["avg_ms"] * 1e-3)
    peak, unit, ach = PEAK_HBM_GBPS, "GB/s", rate / 1e9
    traffic, traffic_src = side_traffic(name, dom)
    return ({
        "metric": "samples/sec fwd+bwd " + name, "value": B * args.steps / dt, "unit": "samples/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 products, f32 accumulate" if f16 else "f32", "data": "synthetic",
        "config": {"workload": name},
        "roofline": {"bound": bound, "kernel": dom, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                     "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": d["avg_ms"],
                     "algorithmic_bytes_per_launch": rate * d["avg_ms"] * 1e-3},
        **extra,
        "kernels": {k: dict(avg_ms=round(v["avg_ms"], 4), work=v["work"]) for k, v in sorted(ks.items())},
        "gpu_kernel_ms_per_step": sum(v["total_ms"] for v in ks.values()) / args.steps,
        "hipgraph_replay_ms_per_step": graph_ms})


def side_workloads(args):
    """The other BASELINE configs, a few steps each, AFTER the headline's timed region and in the same process (so the driver's
    one run timestamps them): c2 DeepFM whole model bf16, c3 DCN cross layers, c5 AutoInt 3-layer f16-MFMA stack, and the
    north-star layer inside its model (xDeepFM).  Compact: ms_per_step, samples/s, the dominant kernel's roofline fraction."""
    import copy
    out = {}
    for name, kw in (("c2_deepfm_bf16", dict(workload="deepfm", graph=True)), ("c3_dcn", dict(workload="dcn", graph=True)),
                     ("c5_autoint_f16_L3", dict(workload="autoint", precision="f16_mfma", layers=3, graph=False)),
                     ("c4_xdeepfm_whole_model", dict(workload="xdeepfm", graph=True))):
        a = copy.copy(args)
        a.steps, a.warmup, a.batch = 10, 3, 0
        for k, v in kw.items():
            setattr(a, k, v)
        if a.workload in ("dcn", "fm"):      # launch-bound (0.05 ms of kernels per step): the eager time is host time, which needs
            a.steps, a.warmup = 200, 50      # more than ten steps to settle (10 steps read 0.06-0.33 ms on one box)
        elif a.workload in ("deepfm", "xdeepfm"):
            a.steps, a.warmup = 30, 10
        elif a.workload == "autoint":        # (its inputs are drawn on the host first: the device idles, and the first ~20 ms after a
            a.steps, a.warmup = 20, 10       # pause run a few % slow -- see the headline's warm-up note)
        try:
            r = deepfm_benchmark(a) if a.workload in ("deepfm", "xdeepfm") else side_benchmark(a)
            e = {"workload": r["config"]["workload"], "ms_per_step": round(r["ms_per_step"], 4), "samples_per_s": round(r["value"], 1),
                 "dtype": r["dtype"], "steps": a.steps}
            if r.get("hipgraph_replay_ms_per_step"):
                e["hipgraph_replay_ms_per_step"] = round(r["hipgraph_replay_ms_per_step"], 4)
                # host-side cost of launching the step op by op: a training user takes the replayed step (layers.capture_step, the
                # default of examples/train_ctr.py)
                e["eager_over_replay"] = round(r["ms_per_step"] / r["hipgraph_replay_ms_per_step"], 2)
            if "roofline" in r:
                e["roofline"] = {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
            if "valu_roofline" in r:
                e["valu_issue_frac"] = r["valu_roofline"]["frac_per_kernel"]
            out[name] = e
        except Exception as exc:     # a side workload must never take the headline line down with it
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    return out


def side_traffic(workload_name, kernel_scope):
    """(HBM bytes per launch, source file) of a side benchmark's kernel from the newest committed profile of the SAME workload
    (profiles/r*_pmc_summary.json written by tools/profile_side.sh / profile_attn.sh next to the bench.json it profiled); a
    lookup, like pmc_traffic().  (None, None) when no committed profile ran this exact workload."""
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    prefix = {"fm_fwd": "fm_fwd", "fm_bwd": "fm_bwd", "dcn_fwd": "dcn_fwd", "dcn_bwd": "dcn_bwd_kernel", "attn_fwd": "attn_fwd",
              "attn_bwd": "attn_bwd"}.get(kernel_scope)
    for f in sorted(glob.glob(os.path.join(root, "r*_pmc_summary.json")), reverse=True):
        bj = f.replace("_pmc_summary.json", "_bench.json")
        try:
            with open(bj) as fh:
                if json.load(fh)["config"]["workload"] != workload_name:
                    continue
            with open(f) as fh:
                per = json.load(fh)
        except (OSError, ValueError, KeyError):
            continue
        hits = [v["hbm_bytes"] for k, v in per.items() if prefix and k.startswith(prefix) and "hbm_bytes" in v]
        if hits:
            return sum(hits) / len(hits), "committed profile " + os.path.basename(f)
    return None, None


def pmc_traffic(scope):
    """(HBM bytes per launch of the kernel behind profiler scope `scope`, name of the file they come from).  Hardware
    counters cannot be read from inside the timed run: the number is a LOOKUP in the newest committed PMC summary
    (profiles/r*_pmc_traffic.json, written by tools/pmc_traffic.py from two separate rocprofv3 --pmc passes of this same
    command by tools/profile_round.sh) and is labelled as such in the JSON line.  (None, None) when no summary matches."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    m = re.match(r"cin_(fwd|bwd_dz|bwd_dw)_(l\d|tail|q)$", scope)
    if not files or not m:
        return None, None
    with open(files[-1]) as fh:
        per = json.load(fh)["per_launch"]
    src = "committed profile " + os.path.basename(files[-1])
    hit = _gemm_launch_of(per, m.group(1), m.group(2))
    return (per[hit]["hbm_bytes"], src) if hit is not None else (None, None)


def _gemm_launch_of(per, kind, layer):
    """Key (in a per-launch PMC summary) of the GEMM launch behind profiler scope cin_<kind>_<layer>.  The exact-fp32 step runs its
    GEMMs in a fixed order: forward l1, l2, .., tail; backward tail, .., l2, l1 -- and a kernel launched several times per step appears
    as '<name> #slot' entries.  The quadratic tail runs the FIRST layer's kernels a second time (forward: second launch; backward:
    first), the fused tail has kernels of its own (cin_tail_*)."""
    if layer == "q":   # merged quadratic tail (cin_qmerge.h): one forward / weight-gradient / data-gradient launch for layer 1 + the quadratic form
        name = {"fwd": "cin_fwdq_kernel", "bwd_dw": "cin_dwq_kernel", "bwd_dz": "cin_dz2_kernel"}.get(kind)
        hits = sorted(k for k in per if name is not None and k.startswith(name))
        return hits[0] if hits else None
    prefixes = {"fwd": ("cin_fwd3_kernel", "cin_tail_fwd_kernel"), "bwd_dz": ("cin_dz3_kernel", "cin_tail_dz_kernel"),
                "bwd_dw": ("cin_dw3_kernel", "cin_tail_dw_kernel")}[kind]
    hits = sorted((v["first_dispatch"], k) for k, v in per.items() if k.startswith(prefixes))
    if not hits:
        return None
    order = [k for _, k in hits] if kind == "fwd" else [k for _, k in hits][::-1]     # -> l1, l2, .., (tail)
    if layer == "tail":
        return order[-1] if len(order) >= 2 else None
    i = int(layer[1:]) - 1
    return order[i] if i < len(order) else None


def mfma_util(scope):
    """MFMA-pipe utilisation of the kernel behind `scope` from the newest committed SQ-counter summary
    (profiles/r*_mfma_util.json, tools/profile_round.sh); a lookup like pmc_traffic().  None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_mfma_util.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        per = json.load(fh)
    e = per.get("by_scope", {}).get(scope)
    if not e:
        return None
    return {"mfma_busy_frac": e.get("mfma_busy_frac"), "util_source": "committed profile " + os.path.basename(files[-1])}


def split_mfma_util(scope):
    """MFMA-pipe utilisation of the split-bf16 kernel behind `scope` from the newest committed SQ-counter summary of a --cin-mode 2
    run (profiles/r*_cin_bf16x3_mfma_util.json, tools/pmc_split.sh); a lookup like mfma_util().  None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_cin_bf16x3_mfma_util.json")))
    name = {"cin_fwd_q": "cin_fwdq_b_kernel", "cin_bwd_dw_q": "cin_dwq_b_kernel", "cin_bwd_dz_q": "cin_dz2_b_kernel"}.get(scope)
    if not files or name is None:
        return None
    with open(files[-1]) as fh:
        per = json.load(fh).get("per_launch", {})
    for k, v in per.items():
        if k.startswith(name):
            return {"mfma_busy_frac": v.get("mfma_busy_frac"), "util_source": "committed profile " + os.path.basename(files[-1])}
    return None


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh children through torch.distributed.run.
    The parent has not touched a GPU (no HIP call, no torch.cuda query) and never will; it relays rank 0's JSON line
    and propagates the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(r.stdout)
    return r.returncode if r.returncode != 0 or lines else 1


class stdout_to_stderr:
    """RCCL prints a version banner on the C-level stdout when the first communicator comes up; the contract of this
    script is ONE JSON line on stdout, so file descriptor 1 points at stderr while the process group initialises."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:   # RCCL prints through C stdio, which is block-buffered when stdout is a file or pipe: flush it while fd 1 still
            import ctypes   # points at stderr, or the banner comes out at exit, behind the JSON line
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def timed_steps(step, fence, steps):
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the compact side_workloads block (configs 2, 3, 5 + the whole model)")
    ap.add_argument("--workload", default="cin", choices=["cin", "fm", "dcn", "autoint", "deepfm", "xdeepfm"],
                    help="cin = the headline benchmark (default); the others are single-GPU side benchmarks of the "
                         "remaining hot-path rows (BASELINE.json configs 2, 3, 5)")
    ap.add_argument("--graph", action="store_true", help="side benchmarks: also time the step replayed from a HIP graph")
    ap.add_argument("--cin-mode", type=int, default=0, help="fil_cin mode bits (experiments; the headline is mode 0)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16_mfma"], help="AutoInt side benchmark only")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of a side benchmark")
    ap.add_argument("--layers", type=int, default=3, help="AutoInt side benchmark: stacked interacting layers (config 5: 3)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: B=4096 per GPU (default); strong: global B=4096 split over the GPUs")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce of the whole bucket after the backward (= --overlap off)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="layer-wise all-reduce on a side stream behind the backward's grad-ready events: on / off / auto = on when "
                         "there is a second rank to exchange with (on one rank the two stream hops cost ~0.02 ms and hide nothing)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the collectives even at world size 1 (plumbing check on one GPU)")
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps: `value` / `ms_per_step` are the FIRST window (the contract's K steps); the others "
                         "only feed ms_per_step_windows / _median / _min / _max, so that a 3 %% kernel change can be told from box noise")
    ap.add_argument("--no-graph-replay", action="store_true", help="skip the HIP-graph replay timing of the headline step (profiling runs: "
                    "keeps the number of step iterations the PMC summarisers expect)")
    ap.add_argument("--no-candidate", action="store_true", help="skip the split-bf16 candidate (FIL_CIN_BF16X3) timed beside the headline")
    ap.add_argument("--stub", default="", help=argparse.SUPPRESS)   # tests: module with install(namespace) -> CPU/gloo stand-ins
    args = ap.parse_args()
    if args.workload in ("deepfm", "xdeepfm"):
        print(json.dumps(deepfm_benchmark(args)))
        return 0
    if args.workload != "cin":
        print(json.dumps(side_benchmark(args)))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.overlap == "off" or (args.overlap == "auto" and world == 1):
        args.no_overlap = True
    import torch.distributed as dist
    from ml_function_amd import dp
    ns = dict(backend="nccl", device=None, Fn=None, profile=True, shape=dict(batch=B_PER_GPU, conv=CONV, fields=F, embed=K))
    if args.stub:   # CPU/gloo stand-ins for the kernels (tests/test_dp_gloo.py drives the launcher and the step logic with them)
        import importlib
        importlib.import_module(args.stub).install(ns)
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        ns["device"] = torch.device("cuda", local_rank)
        from ml_function_amd import functional as Fn
        ns["Fn"] = Fn
    device, Fn = ns["device"], ns["Fn"]
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = dict(device_id=device) if device.type == "cuda" else {}
        with stdout_to_stderr():
            dist.init_process_group(ns["backend"], rank=rank, world_size=world, **kw)
            dist.barrier()      # first collective: the communicator (and RCCL's banner) comes up here
        if dist.get_world_size() != args.gpus:   # the line below claims n_gpus = --gpus: never print it for another group size
            sys.exit("bench.py --gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))

    shape = dict(ns["shape"])
    if args.scaling == "strong":
        lo, hi = dp.shard_bounds(shape["batch"], rank, world)
        shape["batch"] = hi - lo
    inp = make_inputs(rank, device, param_batch=ns["shape"]["batch"], **shape)
    flat, grads, segments = make_bucket(inp, device)
    L = len(inp["Ws"])
    # one collective per POINT of the backward at which gradients become final (fil.h fil_cin_grad_ready_points: with the fused
    # tail the two top layers finish together), each behind the grad_ready slot of the lowest layer of its group
    layer_of_event = list(range(L - 1, -1, -1))
    if device.type == "cuda" and not args.no_overlap:
        sh0 = inp["x"].shape
        points = Fn.cin_grad_ready_points(sh0[0], sh0[1], sh0[2], [int(w.shape[1]) for w in inp["Ws"]], args.cin_mode)
        if use_dist:      # the points depend on the LOCAL batch: an uneven shard can sit on the other side of the library's threshold
            points = dp.agree_on_points(points, device)
        segments, layer_of_event = dp.merge_segments_by_point(segments, points)
    reducer = dp.LayerwiseAllReduce(flat, segments if not args.no_overlap else [(0, flat.numel())], force=args.force_collective)
    ready = None
    if device.type == "cuda" and use_dist and not args.no_overlap:
        ready = [None] * (L + 1)
        for ev, l in zip(reducer.events, layer_of_event):
            ready[l] = ev

    def compute(mode):
        out, pooled, saved = Fn.cin_forward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], inp["dense_b"], 1, mode)
        Fn.cin_backward_raw(inp["x"], inp["Ws"], inp["bs"], inp["dense_w"], pooled, saved, inp["g"], 1, mode, grads=grads,
                            ready_events=ready)
        return out

    def step(mode=args.cin_mode, comm=True):
        out = compute(mode)
        if use_dist and comm:
            if ready is None and device.type == "cuda":   # --no-overlap: the whole bucket after the backward, on the compute stream
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                return out
            reducer.launch()   # sum of layer gradients over the data-parallel ranks (RCCL over xGMI)
            reducer.wait()
        return out

    def fence():
        if use_dist:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()

    per_rank = {}

    def max_over_ranks(dt, tag=None):
        if not use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        if tag is not None:      # every rank's own time of the headline region (the JSON carries min / max / all of them)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank[tag] = [float(e.item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    prof_on = ns["profile"]
    if prof_on:
        from ml_function_amd import _lib
    GEMMS = "cin_fwd_,cin_bwd_dw_,cin_bwd_dz_"     # the MFMA GEMM scopes: _l1 (pair-symmetric first layer), _l2.., _tail, _q (merged quadratic tail)
    for _ in range(args.warmup):
        step()
    # An event pair costs ~5-10 us of stream time (all six GEMM scopes: the step is 5 % slower, all 15 scopes: 7 %), so the timed
    # region carries HIP events around ONE kernel, the dominant one -- picked by an untimed pre-pass with the six GEMM scopes
    # recorded; the per-kernel table of the other GEMMs comes from a second untimed pass in front of the timed region.
    dom_scope, n_pre, prof = None, max(2, args.steps // 4), {}
    if prof_on:
        _lib.profile_begin(GEMMS)
        for _ in range(n_pre):
            step()
        fence()
        pre = _lib.profile_end()
        dom_scope = max(pre, key=lambda k: pre[k]["total_ms"])
        # the per-kernel table of the GEMM scopes: an untimed pass of K steps, run BEFORE the timed region (it used to follow it) -- the
        # device then enters the timed steps after ~30 ms of continuous work instead of ~10 (steady clocks, see below)
        _lib.profile_begin(GEMMS)
        for _ in range(args.steps):
            step()
        fence()
        prof = _lib.profile_end()
        _lib.profile_begin(dom_scope)
    # The W warm-up steps run HERE, directly in front of the K timed ones and under their conditions (the dominant kernel's event pair
    # on): the pre-pass above ends in a device synchronisation plus host-side parsing, and the first steps after such a pause run ~5 %
    # slower (round 5: windows of 20 steps measured 0.783, 0.745, 0.744, 0.744, 0.743 ms per step with the pause in front of the first;
    # 0.759-0.765 with these steps in front and ~10 ms of work before them).  The dominant kernel's average therefore covers W + K launches.
    for _ in range(args.warmup):
        step()
    dt = max_over_ranks(timed_steps(step, fence, args.steps), tag="headline")
    prof_dom = _lib.profile_end() if prof_on else {}
    # further windows of the same K steps under the same conditions (the dominant kernel's event pair stays on): spread only
    window_dts = [dt]
    for _ in range(max(1, args.windows) - 1):
        if prof_on:
            _lib.profile_begin(dom_scope)
        for _ in range(args.warmup):
            step()
        window_dts.append(max_over_ranks(timed_steps(step, fence, args.steps)))
        if prof_on:
            _lib.profile_end()

    # The LABELLED reduced-operand mode of the same step (fil.h FIL_CIN_BF16X3: the three GEMM launches on split-bf16 operands, six
    # bf16 MFMAs per product, fp32 accumulate), timed under the headline's protocol AFTER its windows and reported BESIDE it:
    # `value` / `ms_per_step` above stay the exact-fp32 chain.
    cand = None
    if device.type == "cuda" and not use_dist and not args.stub and not args.no_candidate and (args.cin_mode & 2) == 0:
        try:
            m2 = args.cin_mode | Fn.CIN_BF16X3
            for _ in range(args.warmup):
                step(mode=m2)
            dt2 = timed_steps(lambda: step(mode=m2), fence, args.steps)
            _lib.profile_begin(GEMMS)
            for _ in range(args.steps):
                step(mode=m2)
            fence()
            cand = dict(dt=dt2, prof={k: v for k, v in _lib.profile_end().items() if v.get("executed", 0) > 1e9})
        except Exception as e:   # (must not take the headline line down with it)
            print("bench.py: the split-bf16 candidate failed: %r" % (e,), file=sys.stderr)

    # collective evidence: the same steps without the all-reduce (exposed = difference) and the collectives alone
    rccl = None
    if use_dist:
        dt_nocomm = max_over_ranks(timed_steps(lambda: step(comm=False), fence, args.steps))
        def comm_only():
            if ready is None and device.type == "cuda":
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                return
            if device.type == "cuda":
                for ev in reducer.events:
                    ev.record()
            reducer.launch()
            reducer.wait()
        for _ in range(2):
            comm_only()
        dt_comm = max_over_ranks(timed_steps(comm_only, fence, args.steps))
        rccl = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                "nccl_version": ".".join(map(str, torch.cuda.nccl.version())) if device.type == "cuda" else None,
                "allreduce_bytes": int(flat.numel() * 4), "segments_bytes": [int((b - a) * 4) for a, b in reducer.segments],
                "overlap": "one collective per point of the backward at which gradients become final (top layers first), on a side stream behind fil.h's grad_ready_events" if not args.no_overlap
                           else "none: one all-reduce on the compute stream after the backward",
                "collectives_forced_at_world_size_1": bool(args.force_collective and world == 1),
                "ms_per_step_per_rank": [t / args.steps * 1e3 for t in per_rank.get("headline", [])],
                "ms_per_step_min_rank": min(per_rank["headline"]) / args.steps * 1e3 if per_rank.get("headline") else None,
                "ms_per_step_max_rank": max(per_rank["headline"]) / args.steps * 1e3 if per_rank.get("headline") else None,
                "allreduce_alone_ms": dt_comm / args.steps * 1e3,
                "step_without_allreduce_ms": dt_nocomm / args.steps * 1e3,
                "exposed_allreduce_ms": (dt - dt_nocomm) / args.steps * 1e3}

    # separate, untimed pass with every scope recorded: the per-kernel table of the small kernels
    prof_all, n_all = {}, max(2, args.steps // 4)
    if prof_on:
        _lib.profile_begin(None)
        for _ in range(n_all):
            step()
        fence()
        prof_all = _lib.profile_end()

    # the same step replayed from a HIP graph (the C ABI neither allocates nor synchronises: capturable as it stands).  Reported beside
    # the headline, never as it: `value` stays the eagerly launched step.
    def replay_ms(mode):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                compute(mode)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            compute(mode)
        for _ in range(max(2, args.warmup)):
            gr.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            gr.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / args.steps * 1e3

    graph_ms = None
    if device.type == "cuda" and not use_dist and not args.stub and not args.no_graph_replay:
        try:
            graph_ms = replay_ms(args.cin_mode)
            if cand is not None:      # (the candidate's step is a quarter shorter: on a slow host its eager launches are the first to fall behind)
                cand["graph_ms"] = replay_ms(args.cin_mode | Fn.CIN_BF16X3)
        except Exception as e:   # (a capture problem must not take the headline line down with it)
            print("bench.py: HIP-graph replay of the headline step failed: %r" % (e,), file=sys.stderr)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        global_batch = world * shape["batch"] if args.scaling == "weak" else ns["shape"]["batch"]
        value = global_batch * args.steps / dt
        res = {
            "metric": "samples/sec fwd+bwd xDeepFM-CIN B=4096,F=39,K=16",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "ms_per_step_windows": [w / args.steps * 1e3 for w in window_dts],
            "ms_per_step_median": sorted(window_dts)[len(window_dts) // 2] / args.steps * 1e3,
            "ms_per_step_min": min(window_dts) / args.steps * 1e3, "ms_per_step_max": max(window_dts) / args.steps * 1e3,
            "dtype": "f32", "data": "synthetic",
            "untimed_steps_before_the_timed_ones": {
                "warmup_before_profiling_passes": args.warmup, "dominant_kernel_pick_pass": n_pre if prof_on else 0,
                "per_kernel_table_pass": args.steps if prof_on else 0, "warmup_directly_in_front_of_the_timed_steps": args.warmup,
                "note": "`warmup` above = the W steps directly in front of the K timed ones; the untimed per-kernel passes run before them"},
            "config": {"workload": "xDeepFM CIN 3x128 feature maps fwd+bwd, F=39 K=16, B=%d per GPU, fp32 "
                                   "(BASELINE.json configs[3])" % shape["batch"], "global_batch": global_batch,
                       "parallelism": "dp%d" % world, "grad_allreduce_bytes": int(flat.numel() * 4) if use_dist else 0},
        }
        if prof:
            # dominant kernel = largest total time among the MFMA GEMM scopes.  Every rate below is priced on EXECUTED flops --
            # the products the kernel's algorithm really performs (pair-symmetric first layer: F(F/2+1) of the F^2 channels;
            # fused tail: F+1 instead of H columns; no padding counted) -- so nothing here can exceed the pipe's peak; the
            # algorithmic flops of the reference graph's step that a scope stands for are reported next to them.
            is_gemm = lambda k: k.startswith(("cin_fwd_", "cin_bwd_dw_", "cin_bwd_dz_"))
            mf = {k: v for k, v in prof.items() if is_gemm(k)}
            dom = dom_scope
            d = prof_dom[dom]            # measured inside the timed region
            tf = lambda flops, ms: flops / (ms * 1e-3) / 1e12
            achieved = tf(d["executed"], d["avg_ms"])
            kernels = {}
            for k, v in sorted(prof_all.items()):
                kernels[k] = dict(avg_ms=round(v["avg_ms"], 4), launches_per_step=v["count"] / n_all)
            mf[dom] = d
            # step iterations each scope's record covers: the untimed per-kernel pass ran K steps; the dominant scope's record was
            # open over the W warm-up steps in front of the timed region AND the K timed ones (round 5 divided both by K: the
            # dominant scope read 1.25 launches per step and `executed_frac` came out 8 % high)
            covered = {k: (args.warmup + args.steps) if k == dom else args.steps for k in mf}
            for k, v in mf.items():  # the GEMM kernels: `dom` from the timed region, the others from the untimed pass before it
                kernels[k] = dict(avg_ms=round(v["avg_ms"], 4), launches_per_step=v["count"] / covered[k],
                                  executed_flops_per_launch=v["executed"], executed_tflops=round(tf(v["executed"], v["avg_ms"]), 2),
                                  executed_frac_of_peak=round(tf(v["executed"], v["avg_ms"]) / PEAK_F32_MFMA_TFLOPS, 4),
                                  algorithmic_flops_per_launch=v["work"])
            traffic, traffic_src = pmc_traffic(dom)
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                               "traffic_source": traffic_src, "avg_launch_ms": d["avg_ms"], "flops_per_launch": d["executed"],
                               "flops_are": "executed by the kernel (exact fp32 MFMA products, padding not counted); the reference "
                                            "graph spends algorithmic_flops_per_launch on the same step",
                               "algorithmic_flops_per_launch": d["work"]}
            util = mfma_util(dom)
            if util is not None:
                res["roofline"].update(util)
            if traffic:   # BASELINE's "%HBM roofline" of the same kernel: counter bytes over the live launch time
                res["roofline"]["hbm_frac"] = traffic / (d["avg_ms"] * 1e-3) / (PEAK_HBM_GBPS * 1e9)
            res["kernels"] = kernels
            res["gpu_kernel_ms_per_step"] = sum(v["total_ms"] for v in prof_all.values()) / n_all
            # whole step: executed MFMA flops of all GEMM scopes over the step time (the small VALU kernels add ~1 %)
            exe_step = sum(v["executed"] * v["count"] / covered[k] for k, v in mf.items())
            sh = shape
            algo_step = 3.0 * 2.0 * sh["embed"] * sh["batch"] * sum(
                (sh["fields"] if l == 0 else sh["conv"][l - 1]) * sh["fields"] * h for l, h in enumerate(sh["conv"]))
            res["executed_flops_per_step"] = exe_step
            res["algorithmic_flops_per_step"] = algo_step
            res["executed_frac"] = exe_step / (ms_per_step * 1e-3) / (PEAK_F32_MFMA_TFLOPS * 1e12)
        if cand is not None:
            c_ms = cand["dt"] / args.steps * 1e3
            ck = {k: dict(avg_ms=round(v["avg_ms"], 4), executed_flops_per_launch=v["executed"], bf16_mfma_flops_per_launch=6.0 * v["executed"],
                          bf16_tflops=round(6.0 * v["executed"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                          frac_of_bf16_peak=round(6.0 * v["executed"] / (v["avg_ms"] * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4))
                  for k, v in sorted(cand["prof"].items())}
            cd = max(ck, key=lambda k: ck[k]["avg_ms"]) if ck else None
            res["candidate_bf16x3"] = {
                "what": "the same step with fil.h FIL_CIN_BF16X3: the three GEMM launches on split-bf16 operands (each fp32 operand as three "
                        "bf16 pieces, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate); holds the parity bars of the exact mode "
                        "(tests/test_gpu_parity.py, same tests, same bars); reported beside the headline, never as `value`",
                "ms_per_step": c_ms, "samples_per_s": shape["batch"] * args.steps / cand["dt"], "steps": args.steps, "warmup": args.warmup,
                "dtype": "bf16x3 (three bf16 pieces per fp32 operand, fp32 accumulate)", "speedup_over_exact": ms_per_step / c_ms,
                "kernels": ck}
            if cand.get("graph_ms") is not None:
                res["candidate_bf16x3"]["hipgraph_replay_ms_per_step"] = cand["graph_ms"]
            if cd is not None:
                util = split_mfma_util(cd)
                res["candidate_bf16x3"]["roofline"] = {
                    "bound": "mfma", "kernel": cd, "achieved": ck[cd]["bf16_tflops"], "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ck[cd]["frac_of_bf16_peak"], "avg_launch_ms": ck[cd]["avg_ms"],
                    "flops_are": "6 x the executed fp32-equivalent flops (six bf16 MFMAs per product), padding not counted",
                    **(util or {})}
        if graph_ms is not None:
            res["hipgraph_replay_ms_per_step"] = graph_ms
        if rccl is not None:
            res["rccl"] = rccl
        if world == 1 and not args.stub and not args.no_side:
            res["side_workloads"] = side_workloads(args)
        if not args.no_cpu_baseline and world == 1 and not args.stub:  # reported at N=1 only (rank 0)
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if ns.get("on_done") is not None:   # tests: hand out the bucket of one clean step (the comm-only timing passes re-reduce it)
        step()
        fence()
        ns["on_done"](flat=flat, rank=rank, world=world)
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
