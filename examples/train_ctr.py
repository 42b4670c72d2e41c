"""Train one of the re-hosted CTR models on synthetic Criteo-shaped data: the counterpart of the reference's
example/ctr_example/un_seq.py (Adam + binary cross-entropy + AUC, :55-66), on the HIP layers.

  python examples/train_ctr.py --model XDeepFM --steps 200
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ctr.py --model XDeepFM

Data parallel: every rank draws its own shard of each batch, dense gradients go through one bucketed all-reduce
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
from ml_function_amd import dp, metrics, models  # noqa: E402


def make_batch(rng, vocab, n_dense, B, teacher, device):
    idx = np.stack([np.minimum(rng.zipf(1.3, B) - 1, v - 1) for v in vocab], 1)
    dense = rng.random((B, n_dense), dtype=np.float32)
    logit = sum(teacher[f][idx[:, f]] for f in range(len(vocab))) + dense @ teacher["dense"]
    y = (rng.random(B) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return (torch.tensor(dense, device=device), torch.tensor(idx, device=device), torch.tensor(y, device=device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="XDeepFM", choices=["FM", "DeepFM", "DCN", "XDeepFM", "AutoInt", "NFM", "AFM"])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--dense", type=int, default=13)
    ap.add_argument("--embed-dim", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-3)
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
    info = models.make_sparse_info(vocab, embed_dim=args.embed_dim)
    single = args.model == "XDeepFM"
    fi = models.FeatureInput(sparseInfo=info, useLinear=args.model != "DCN" and args.model != "AutoInt", useAddLinear=single,
                             useFlattenLinear=True)
    body = {"FM": models.FM, "DeepFM": models.DeepFM, "DCN": models.DCN, "AutoInt": models.AutoInt, "NFM": models.NFM,
            "AFM": models.AFM, "XDeepFM": lambda: models.XDeepFM(conv_size=[128, 128, 128])}[args.model]()
    torch.manual_seed(0)  # identical replicas
    model = models.CTRModel(fi, body).to(device)
    rng = np.random.default_rng(1000 + rank)
    dense, idx, y = make_batch(rng, vocab, args.dense, args.batch, teacher, device)
    use_dense = args.model not in ("FM", "AutoInt", "AFM")
    model(dense if use_dense else None, idx)  # builds the lazily created weights
    tables = [p for n, p in model.named_parameters() if n.endswith("embeddings")]
    others = [p for n, p in model.named_parameters() if not n.endswith("embeddings")]
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    for step in range(args.steps):
        dense, idx, y = make_batch(rng, vocab, args.dense, args.batch, teacher, device)
        opt.zero_grad(set_to_none=True)
        out = model(dense if use_dense else None, idx)
        p = (out[:, 1] if out.shape[1] == 2 else out[:, 0]).clamp(1e-6, 1 - 1e-6)
        loss = torch.nn.functional.binary_cross_entropy(p, y) / world
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
        opt.step()
        if rank == 0 and (step % 20 == 0 or step == args.steps - 1):
            print("step %4d  loss %.4f  auc %.4f" % (step, float(loss.detach()) * world, metrics.auc(y, p.detach())), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
