"""Train one of the re-hosted CTR models on synthetic Criteo-shaped data: the counterpart of the reference's
example/ctr_example/un_seq.py (Adam + binary cross-entropy + AUC, :55-66), on the HIP layers.

  python examples/train_ctr.py --model XDeepFM --steps 200
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ctr.py --model XDeepFM

The input side is the reference's too: the raw click log is a CSV-shaped pandas frame (string categories with missing values, dense
columns with NaNs) that goes through the field-index front end ml_function_amd.data_prepare (sparse_fea_deal: fillna('-1') ->
astype(str) -> label encoding, one sparseFea per column; dense_fea_deal: fillna(mode) -> min-max scaling; kon/utils/data_prepare.py:
85-100, 294-301); then (:335-337) the encoded table is sliced, shuffled with a
2048-element buffer, repeated, batched and prefetched by ml_function_amd.data.data_pipeline (host -> device copies on a
side stream); the loss is Keras' compiled loss: binary cross-entropy + the layers' regularisation terms (the l2(emb_reg)
of every embedding table, interactive_layer.py:217); the optimiser is Keras' 'adam' (lr 1e-3, epsilon 1e-7).
Data parallel: every rank trains on its own shard of the table, dense gradients go through one bucketed all-reduce
(ml_function_amd.dp.allreduce_module_grads), the embedding tables exchange only the rows their shards touched
(dp.exchange_sparse_rows).  Labels come from a fixed random "teacher" so that the AUC has something to learn.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd  # noqa: E402

from ml_function_amd import data, dp, losses, metrics, models  # noqa: E402
from ml_function_amd.data_prepare import data_prepare  # noqa: E402
from ml_function_amd.layers.base import collect_regularization_loss  # noqa: E402


def make_raw_log(rng, vocab, n_dense, rows, teacher):
    """A synthetic click log as it would come out of a CSV: (sparse frame of STRING categories with ~2 % missing values, dense frame
    of floats with ~2 % NaNs, label [rows]).  The labels follow a fixed random teacher on the raw values."""
    idx = np.stack([np.minimum(rng.zipf(1.3, rows) - 1, v - 1) for v in vocab], 1)
    dense = rng.random((rows, n_dense)) * 100.0
    logit = sum(teacher[f][idx[:, f]] for f in range(len(vocab))) + (dense / 100.0) @ teacher["dense"]
    y = (rng.random(rows) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    sparse = pd.DataFrame({"C%d" % (f + 1): np.where(rng.random(rows) < 0.02, None, np.char.add("v", idx[:, f].astype("U"))) for f in range(len(vocab))})
    dense_df = pd.DataFrame({"I%d" % (i + 1): np.where(rng.random(rows) < 0.02, np.nan, dense[:, i]) for i in range(n_dense)})
    return sparse, dense_df, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="XDeepFM", choices=["FM", "DeepFM", "DCN", "XDeepFM", "AutoInt", "NFM", "AFM"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--dense", type=int, default=13)
    ap.add_argument("--embed-dim", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--no-graph", action="store_true",
                    help="run every step eagerly (default on one GPU: the whole step -- forward, backward, Adam -- is captured once "
                         "into a HIP graph and replayed; the C ABI neither allocates nor synchronises, so it is capture-safe)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    rng0 = np.random.default_rng(2020)
    vocab = [int(v) for v in np.exp(rng0.uniform(np.log(10), np.log(2e5), args.fields))]
    teacher = {f: rng0.normal(0, 0.5, v).astype(np.float32) for f, v in enumerate(vocab)}
    teacher["dense"] = rng0.normal(0, 0.5, args.dense).astype(np.float32)
    # raw log -> field-index front end -> ids + descriptors.  LabelEncoder ids are ranks in the sorted set of the values PRESENT, so
    # the encoder must see the same log on every rank or one raw category would land in different embedding rows on different
    # replicas (and the averaged gradients would mix unrelated categories): the WHOLE log is drawn from one common seed and encoded
    # identically everywhere; a rank then keeps its own row shard of the encoded frame.
    per_rank = args.steps * args.batch // 2                                                  # repeat(2) makes `steps` batches
    raw_sparse, raw_dense, labels = make_raw_log(np.random.default_rng(1000), vocab, args.dense, per_rank * world, teacher)
    prep = data_prepare(batch_size=args.batch)
    ids_df, info = prep.sparse_fea_deal(raw_sparse, embed_dim=args.embed_dim)
    dense_df, _ = prep.dense_fea_deal(raw_dense)
    lo, hi = rank * per_rank, (rank + 1) * per_rank
    ids_df, dense_df, labels = ids_df.iloc[lo:hi], dense_df.iloc[lo:hi], labels[lo:hi]
    single = args.model == "XDeepFM"
    fi = models.FeatureInput(sparseInfo=info, useLinear=args.model != "DCN" and args.model != "AutoInt", useAddLinear=single,
                             useFlattenLinear=True)
    body = {"FM": models.FM, "DeepFM": models.DeepFM, "DCN": models.DCN, "AutoInt": models.AutoInt, "NFM": models.NFM,
            "AFM": models.AFM, "XDeepFM": lambda: models.XDeepFM(conv_size=[128, 128, 128])}[args.model]()
    torch.manual_seed(0)  # identical replicas
    model = models.CTRModel(fi, body).to(device)
    table = (dense_df.to_numpy(np.float32), ids_df.to_numpy(np.int64), labels)
    use_dense = args.model not in ("FM", "AutoInt", "AFM")
    model(torch.tensor(table[0][:args.batch], device=device) if use_dense else None,
          torch.tensor(table[1][:args.batch], device=device))  # builds the lazily created weights
    tables = [p for n, p in model.named_parameters() if n.endswith("embeddings")]
    # l2(emb_reg) of the tables: added analytically after the sparse exchange (see dp.add_table_l2_grad_), identically per replica
    table_l2 = {id(m.embeddings): m.table_l2_ranges() for m in model.modules() if hasattr(m, "table_l2_ranges") and m.built}
    others = [p for n, p in model.named_parameters() if not n.endswith("embeddings")]
    use_graph = world == 1 and not args.no_graph
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, eps=1e-7, capturable=use_graph)      # Keras 'adam' (un_seq.py:61)
    pipe = data.data_pipeline(table, batch_size=args.batch, shuffle_buffer=2048, repeat=2, prefetch=2, seed=rank, device=device)

    def train_step(dense, idx, y):
        """forward + loss + backward + (data-parallel exchange) + Adam; returns (bce, p) of the batch"""
        opt.zero_grad(set_to_none=True)
        out = model(dense if use_dense else None, idx)
        p = out[:, 1] if out.shape[1] == 2 else out[:, 0]
        bce = losses.binary_crossentropy(p, y, eps=1e-6)       # clip + BCE + mean: one launch (ml_function_amd/losses.py)
        loss = (bce + collect_regularization_loss(model, skip_tables=True)) / world
        loss.backward()
        if world > 1:
            bucket = dp.GradBucket.for_params(others)
            bucket.copy_from_grads(others)
            bucket.all_reduce()
            bucket.assign_to_grads(others)
            offs = fi.sparse_embed.offsets
            rows = (idx + offs).reshape(-1)
            for t in tables:
                if t.grad is not None:
                    dp.exchange_sparse_rows(t.grad, rows)
        for t in tables:
            if t.grad is not None and table_l2.get(id(t)):
                dp.add_table_l2_grad_(t.grad, t.detach(), table_l2[id(t)])
        opt.step()
        return bce.detach(), p.detach()

    graph, static = None, None
    for step, (dense, idx, y) in enumerate(pipe):
        full = idx.shape[0] == args.batch            # (the last batch of the pipeline can be short: it runs eagerly)
        if use_graph and full:
            if graph is None:
                # static input buffers; three eager steps on a side stream (lazy initialisations, allocator warm-up), then capture
                static = [torch.empty_like(dense), torch.empty_like(idx), torch.empty_like(y)]
                for dst, src in zip(static, (dense, idx, y)):
                    dst.copy_(src)
                # (the warm-up steps are real optimizer steps: model and Adam state are put back afterwards, so the captured run follows
                # the eager one step for step)
                saved_model = {k: v.clone() for k, v in model.state_dict().items()}
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        train_step(*static)
                torch.cuda.current_stream().wait_stream(side)
                model.load_state_dict(saved_model)
                for st_ in opt.state.values():      # Adam's moments and step count back to "never stepped" (in place: the capture keeps these tensors)
                    for v in st_.values():
                        if torch.is_tensor(v):
                            v.zero_()
                graph = torch.cuda.CUDAGraph()
                opt.zero_grad(set_to_none=True)
                with torch.cuda.graph(graph):
                    static_out = train_step(*static)
            for dst, src in zip(static, (dense, idx, y)):
                dst.copy_(src)
            graph.replay()
            bce, p = static_out
        else:
            bce, p = train_step(dense, idx, y)
        if rank == 0 and (step % 20 == 0 or step == len(pipe) - 1):
            print("step %4d  loss %.4f  auc %.4f" % (step, float(bce), metrics.auc(y, p)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
