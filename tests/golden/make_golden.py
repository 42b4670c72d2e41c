#!/usr/bin/env python
"""Generates the committed golden fixtures under tests/golden/ (run from the repo root, CPU only).

  * fm/dcn/cin/attn_*.npz : seeded inputs + weights and the fp64 outputs/gradients of the oracle's op-for-op
    restatement of the reference graph (oracle/graph.py, autograd in float64).  The reference itself cannot run here
    (TensorFlow is not installable; PARITY UNPINNED, see oracle/__init__.py), so these vectors pin the oracle against
    regressions and give the GPU tests a fixed target; they are not outputs of the reference.
  * label_encode.npz : field-index goldens produced by the REAL third-party code the reference calls --
    sklearn.preprocessing.LabelEncoder on fillna('-1').astype(str) columns (data_prepare.py:91-93) -- plus the
    embedding rows gathered with those indices.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ml_function_amd import synth  # noqa: E402
from oracle import graph  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)


def save(name, **arrays):
    # fp64 oracle results are stored as float32 (2^-24 relative: far below the 1e-5 parity tolerance) to keep fixtures small
    arrays = {k: (np.asarray(v, np.float32) if np.asarray(v).dtype == np.float64 else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name), **arrays)
    print(name, {k: np.asarray(v).shape for k, v in arrays.items()})


def fm(tag, B, F, K, dist):
    c = synth.fm_case(B, F, K, dist=dist)
    emb, lin = T(c["emb"]), T(c["lin"])
    out = graph.fm_layer([emb[:, f:f + 1, :] for f in range(F)], [lin[:, f:f + 1, None] for f in range(F)])
    out.backward(torch.tensor(c["g"], dtype=torch.float64)[:, None, :])
    save("fm_%s.npz" % tag, emb=c["emb"], lin=c["lin"], g=c["g"], out=out.detach().numpy(), demb=emb.grad.numpy(),
         dlin=lin.grad.numpy())


def dcn(tag, B, D, L):
    c = synth.dcn_case(B, D, L)
    x = T(c["x"])
    ws = [T(c["w"][l][:, None]) for l in range(L)]
    bs = [T(c["b"][l][:, None]) for l in range(L)]
    y = graph.cross_layer(x, ws, bs)
    y.backward(torch.tensor(c["g"], dtype=torch.float64)[..., None])
    save("dcn_%s.npz" % tag, x=c["x"], w=c["w"], b=c["b"], g=c["g"], y=y.detach().numpy(), dx=x.grad.numpy(),
         dw=np.stack([w.grad.numpy()[:, 0] for w in ws]), db=np.stack([b.grad.numpy()[:, 0] for b in bs]))


def cin(tag, B, F, K, conv):
    c = synth.cin_case(B, F, K, conv, dist="uniform")
    c["x"] = (c["x"] * 10).astype(np.float32)
    x, Ws, bs = T(c["x"]), [T(w) for w in c["Ws"]], [T(b) for b in c["bs"]]
    dw, db = T(c["dense_w"]), T(c["dense_b"])
    out = graph.cin(x, Ws, bs, dw, db)
    out.backward(torch.tensor(c["g"], dtype=torch.float64))
    arrays = dict(x=c["x"], dense_w=c["dense_w"], dense_b=c["dense_b"], g=c["g"], out=out.detach().numpy(),
                  dx=x.grad.numpy(), ddense_w=dw.grad.numpy(), ddense_b=db.grad.numpy(), conv=np.asarray(conv))
    for l in range(len(conv)):
        arrays["W%d" % l], arrays["b%d" % l] = c["Ws"][l], c["bs"][l]
        arrays["dW%d" % l], arrays["db%d" % l] = Ws[l].grad.numpy(), bs[l].grad.numpy()
    save("cin_%s.npz" % tag, **arrays)


def attn(tag, B, F, K, H, A):
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    names = ["x", "Wq", "Wk", "Wr", "gamma", "beta"]
    t = {n: T(c[n]) for n in names}
    y = graph.autoint_interacting(t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"])
    y.backward(torch.tensor(c["dy"], dtype=torch.float64))
    arrays = {n: c[n] for n in names}
    arrays.update(dy=c["dy"], y=y.detach().numpy(), flat=graph.autoint_flatten(y.detach()).numpy())
    arrays.update({"d" + n: t[n].grad.numpy() for n in names})
    save("attn_%s.npz" % tag, **arrays)


def attn_stack(tag, B, F, K, H, A, L, beta_shift=0.0):
    """BASELINE config 5 as written (3 stacked interacting layers; the stacking rule is the documented extension of
    oracle/graph.py:autoint_stack -- head-concat per the reference's ESULayer, behavior_layer.py:973).
    beta_shift: the kink-free variant (synth.attn_stack_case) the f16 gradient bars are held on."""
    c = synth.attn_stack_case(B, F, K, H, A, L, dist="normal", beta_shift=beta_shift, center_upper=beta_shift > 0)
    x = T(c["x"])
    layers = [tuple(T(p) for p in lay) for lay in c["layers"]]
    y = graph.autoint_stack(x, layers)
    y.backward(torch.tensor(c["dy"], dtype=torch.float64))
    arrays = dict(x=c["x"], dy=c["dy"], y=y.detach().numpy(), dx=x.grad.numpy(), L=np.asarray(L))
    for l, lay in enumerate(layers):
        for n, p, raw in zip(["Wq", "Wk", "Wr", "gamma", "beta"], lay, c["layers"][l]):
            arrays["%s%d" % (n, l)] = raw
            arrays["d%s%d" % (n, l)] = p.grad.numpy()
    save("attn_stack_%s.npz" % tag, **arrays)


def label_encode():
    """Index goldens from the REAL scikit-learn / pandas (the third-party code the reference calls, data_prepare.py:85-100): raw columns
    as the product's front end receives them (raw_*: object arrays, None = missing), their str form, the encoded ids, vocabulary sizes."""
    import pandas as pd
    from sklearn.preprocessing import LabelEncoder
    rng = np.random.default_rng(synth.SEED)
    cols, enc, rawout = {}, {}, {}
    raw = {
        "C1": rng.integers(0, 30, 64).astype(object),               # ints: '10' sorts before '2'
        "C2": rng.choice(["a", "B", "ab", "", "zz", "Ab"], 64).astype(object),
        "C3": np.where(rng.random(64) < 0.25, None, rng.integers(100, 120, 64).astype(object)),  # missing -> '-1'
        "C4": np.where(rng.random(64) < 0.3, np.nan, rng.integers(0, 12, 64).astype(np.float64)),   # float column with NaN: '3.0', '10.0' < '2.0'
        "C5": np.asarray([[7, "7", 7.5, None, "x", -1, "-1", True][i % 8] for i in rng.integers(0, 8, 64)], dtype=object),  # mixed types
    }
    for name, col in raw.items():
        s = pd.Series(col).fillna("-1").astype("str")       # data_prepare.py:91-92
        le = LabelEncoder()
        enc[name] = le.fit_transform(s).astype(np.int64)     # :93
        cols[name] = s.to_numpy().astype("U")
        # the raw column in a form np.savez can hold without pickling: its repr tokens (n = None, f:<float>, i:<int>, b:<bool>, s:<str>)
        rawout[name] = np.asarray(["n" if v is None else ("f:%r" % float(v) if isinstance(v, float) else ("b:%d" % v if isinstance(v, bool) else
                                   ("i:%d" % v if isinstance(v, (int, np.integer)) else "s:" + str(v)))) for v in col]).astype("U")
    vocab = [int(enc[n].max()) + 1 for n in raw]
    tables = [rng.standard_normal((v, 8)).astype(np.float32) for v in vocab]
    idx = np.stack([enc[n] for n in raw], 1)
    gathered = np.stack([tables[f][idx[:, f]] for f in range(len(vocab))], 1)
    save("label_encode.npz", idx=idx, gathered=gathered, vocab=np.asarray(vocab), names=np.asarray(list(raw)).astype("U"),
         table=np.concatenate(tables, 0), **{"col_" + n: cols[n] for n in raw}, **{"raw_" + n: rawout[n] for n in raw})


def dense_minmax():
    """Dense-column goldens from the real pandas / scikit-learn (data_prepare.py:294-301): fillna(mode) then MinMaxScaler(0, 1)."""
    import pandas as pd
    from sklearn.preprocessing import MinMaxScaler
    rng = np.random.default_rng(synth.SEED + 1)
    n = 48
    df = pd.DataFrame({
        "I1": rng.integers(0, 50, n).astype(np.float64),
        "I2": rng.standard_normal(n) * 1e3 + 5e4,
        "I3": np.full(n, 3.25),                                               # constant column: scale 1
        "I4": np.where(rng.random(n) < 0.2, np.nan, rng.integers(0, 6, n).astype(np.float64)),   # missing -> the (smallest) mode
        "I5": rng.integers(-5, 5, n).astype(np.float64) * 1e-9,
    })
    filled = pd.DataFrame({fea: df[fea].fillna(df[fea].mode()[0]) for fea in df})
    out = MinMaxScaler(feature_range=(0, 1)).fit_transform(filled)
    # (float64 kept: this fixture is compared bit for bit, not to a tolerance)
    np.savez_compressed(os.path.join(OUT, "dense_minmax.npz"), raw=df.to_numpy(np.float64), names=np.asarray(list(df)).astype("U"),
                        out=np.asarray(out, np.float64))


if __name__ == "__main__":
    fm("c1_small", 8, 39, 8, "uniform")       # BASELINE config 1 shape (F=39, K=8), small batch
    fm("tiny", 2, 3, 4, "normal")
    dcn("small", 8, 1248, 3)                  # config 3 width
    dcn("tiny", 2, 5, 1)
    cin("c4_narrow", 4, 39, 16, [16, 16, 16])  # north-star field/embedding shape, 3 layers, narrow maps (fixture size)
    cin("tiny", 2, 5, 8, [6, 7])
    attn("c5_small", 2, 200, 16, 4, 16)       # config 5 layer shape, small batch
    attn("default", 2, 39, 16, 3, 8)          # reference defaults: attention_dim=8, 3 heads
    attn_stack("c5_small", 2, 200, 16, 4, 16, 3)   # config 5: 3 layers, 4 heads, F=200, K=16, A=16, small batch
    attn_stack("c5_nokink", 2, 200, 16, 4, 16, 3, beta_shift=4.0)   # ... with every output away from the ReLU kink
    label_encode()
    dense_minmax()
