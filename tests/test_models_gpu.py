"""N2/N4: whole models assembled from the re-hosted layers run on the HIP path, match the oracle graph composed the
same way, and train (Adam + binary cross-entropy, as example/ctr_example/un_seq.py:61-62 does)."""
import numpy as np
import pytest
import torch

from ml_function_amd import models
from oracle import graph

pytestmark = pytest.mark.gpu


def _inputs(B, n_dense, vocab, seed=0):
    rng = np.random.default_rng(seed)
    dense = torch.tensor(rng.random((B, n_dense)), dtype=torch.float32, device="cuda")
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in vocab], 1), device="cuda")
    return dense, idx


def rel(a, b):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_xdeepfm_matches_oracle_composition():
    vocab = [7, 11, 5, 13, 3, 17]
    B, K = 33, 8
    info = models.make_sparse_info(vocab, embed_dim=K)
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useAddLinear=True, useFlattenLinear=True)
    model = models.CTRModel(fi, models.XDeepFM(conv_size=[16, 12], hidden_units=[32, 16])).cuda()
    dense, idx = _inputs(B, 3, vocab)
    out = model(dense, idx)
    assert out.shape == (B, 1) and float(out.detach().min()) > 0 and float(out.detach().max()) < 1
    # oracle: same parameters, reference graph in float64 on the CPU
    D = lambda t: t.detach().cpu().double()
    emb = fi.sparse_embed.embeddings
    offs = fi.sparse_embed.offsets.cpu()
    tables = [D(emb)[offs[f]:offs[f] + vocab[f]] for f in range(len(vocab))]
    sparse_embed = graph.sparse_embed(tables, [idx[:, f:f + 1].cpu() for f in range(len(vocab))])
    lin_tab = D(fi.linear_embed.embeddings)
    loff = fi.linear_embed.offsets.cpu()
    linear = sum(lin_tab[loff[f]:loff[f] + vocab[f]][idx[:, f].cpu()] for f in range(len(vocab)))  # use_add, flattened [B,1]
    body = model.body
    cin_out = graph.cin(torch.cat(sparse_embed, 1), [D(w)[0] for w in body.cin.conv_kernels], [D(b) for b in body.cin.conv_biases],
                        D(body.cin.logit_kernel), D(body.cin.logit_bias))
    x = graph.stack_layer([D(dense)[:, i:i + 1] for i in range(3)] + sparse_embed)
    for h in body.dnn.hidden_list:
        y = x @ D(h.dense.kernel) + D(h.dense.bias)
        x = torch.relu(x + y) if x.shape == y.shape else torch.relu(y)
    dnn_out = x @ D(body.dnn.logit_layer.kernel) + D(body.dnn.logit_layer.bias)
    want = torch.sigmoid(linear + cin_out + dnn_out)
    assert rel(out, want) < 1e-5


D64 = lambda t: t.detach().cpu().double()


def _oracle_features(fi, vocab, dense, idx, round_bf16=False):
    """The reference's InputFeature pieces in float64 from the model's own tables (oracle/graph.py:sparse_embed)."""
    emb, offs = D64(fi.sparse_embed.embeddings), fi.sparse_embed.offsets.cpu()
    if round_bf16:
        emb = emb.float().bfloat16().double()
    sparse = graph.sparse_embed([emb[offs[f]:offs[f] + vocab[f]] for f in range(len(vocab))],
                                [idx[:, f:f + 1].cpu() for f in range(len(vocab))])          # F x [B,1,K]
    linear = None
    if fi.linear_embed is not None:
        lin, loff = D64(fi.linear_embed.embeddings), fi.linear_embed.offsets.cpu()
        linear = [lin[loff[f]:loff[f] + vocab[f]][idx[:, f].cpu()].unsqueeze(1) for f in range(len(vocab))]   # F x [B,1,1]
    dense_l = [] if dense is None else [D64(dense)[:, i:i + 1] for i in range(dense.shape[1])]
    return sparse, linear, dense_l


def _oracle_dnn(dnn, x):
    """DnnLayer(res_unit=1) with plain Dense hidden layers (core_layer.py:201-226): y = Dense(x); x = ReLU(Add([x, y])) when
    the shapes agree, ReLU(y) otherwise (the Add raises and is skipped, :211-214)."""
    for h in dnn.hidden_list:
        y = x @ D64(h.dense.kernel) + D64(h.dense.bias)
        x = torch.relu(x + y) if x.shape == y.shape else torch.relu(y)
    return x


def _oracle_merge_score(score, inputs, use_merge=True):
    x = graph.stack_layer(inputs) if use_merge else inputs
    return torch.softmax(x @ D64(score.dense.kernel) + D64(score.dense.bias), dim=-1)


@pytest.mark.parametrize("name", ["FM", "DeepFM", "DCN", "AutoInt"])
def test_zoo_matches_oracle_composition(name):
    """Whole zoo models (embedding gather + interaction layer on the HIP path + heads) against the reference graph composed
    the same way in float64 (models.py:36-41, 80-106, 150-165), outputs at 1e-5 and the embedding-table gradient at 2e-5."""
    torch.manual_seed(1)
    vocab = [7, 11, 5, 13, 3, 17]
    B, K = 33, 8
    fi = models.FeatureInput(sparseInfo=models.make_sparse_info(vocab, embed_dim=K), useLinear=name in ("FM", "DeepFM"))
    body = {"FM": lambda: models.FM(), "DeepFM": lambda: models.DeepFM(hidden_units=[32, 16]),
            "DCN": lambda: models.DCN(hidden_units=[32, 16], cross_hidden=3),
            "AutoInt": lambda: models.AutoInt(attention_dim=8, attention_head_dim=3)}[name]()
    model = models.CTRModel(fi, body).cuda()
    dense, idx = _inputs(B, 3, vocab, seed=4)
    if name in ("FM", "AutoInt"):
        dense = None
    out = model(dense, idx)
    sparse, linear, dense_l = _oracle_features(fi, vocab, dense, idx)
    emb64 = D64(fi.sparse_embed.embeddings).requires_grad_()
    offs = fi.sparse_embed.offsets.cpu()
    sparse = graph.sparse_embed([emb64[offs[f]:offs[f] + vocab[f]] for f in range(len(vocab))], [idx[:, f:f + 1].cpu() for f in range(len(vocab))])
    b = model.body
    if name == "FM":
        want = _oracle_merge_score(b.score, graph.fm_layer(sparse, linear).squeeze(1), use_merge=False)
    elif name == "DeepFM":
        dnn_ = _oracle_dnn(b.dnn, graph.stack_layer(dense_l + sparse))
        want = _oracle_merge_score(b.score, [graph.fm_layer(sparse, linear), dnn_])
    elif name == "DCN":
        comb = graph.stack_layer(dense_l + sparse)
        L = b.cross.cross_hidden
        cross = graph.cross_layer(comb, [D64(getattr(b.cross, "outer_weight_%d" % l)) for l in range(L)],
                                  [D64(getattr(b.cross, "outer_bias_%d" % l)) for l in range(L)])
        want = _oracle_merge_score(b.score, [cross, _oracle_dnn(b.deep, comb)])
    else:
        a = b.atten_layer
        y = graph.autoint_interacting(torch.cat(sparse, 1), D64(a.query_w), D64(a.key_w), D64(a.res_w), D64(a.ln_gamma), D64(a.ln_beta))
        want = _oracle_merge_score(b.score, graph.autoint_flatten(y), use_merge=False)
    assert out.shape == want.shape == (B, 2) and rel(out, want) < 1e-5, rel(out, want)
    g = torch.tensor(np.random.default_rng(8).standard_normal((B, 2)), dtype=torch.float64)
    want.backward(g)
    out.backward(g.float().cuda())
    assert rel(fi.sparse_embed.embeddings.grad, emb64.grad) < 2e-5


def test_xdeepfm_adam_step_matches_oracle():
    """N4: ONE training step as the reference runs it (un_seq.py:61: loss = binary cross-entropy + sum(model.losses), here
    the l2(emb_reg) of the embedding tables; optimizer 'adam' = lr 1e-3, beta 0.9/0.999, epsilon 1e-7) on the HIP model,
    against the same step on the float64 oracle graph with a hand-written Adam update: every parameter AFTER the update
    within 1e-5 (norm-relative), the update itself (new - old) within 1e-3."""
    from ml_function_amd.layers.base import collect_regularization_loss
    torch.manual_seed(2)
    vocab = [7, 11, 5, 13, 3, 17]
    B, K = 48, 8
    info = [i._replace(emb_reg=1e-3) for i in models.make_sparse_info(vocab, embed_dim=K)]
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useAddLinear=True, useFlattenLinear=True)
    model = models.CTRModel(fi, models.XDeepFM(conv_size=[16, 12], hidden_units=[32, 16])).cuda()
    dense, idx = _inputs(B, 3, vocab, seed=9)
    y = torch.tensor(np.random.default_rng(10).integers(0, 2, B), dtype=torch.float32, device="cuda")
    model(dense, idx)                                        # lazy build
    names = [n for n, _ in model.named_parameters()]
    before = {n: p.detach().cpu().double().clone() for n, p in model.named_parameters()}
    # ---- oracle: the reference graph in float64 on copies of the parameters
    P = {n: v.clone().requires_grad_() for n, v in before.items()}
    b = model.body
    key = {id(p): n for n, p in model.named_parameters()}
    O = lambda p: P[key[id(p)]]
    offs, loff = fi.sparse_embed.offsets.cpu(), fi.linear_embed.offsets.cpu()
    emb, lin = O(fi.sparse_embed.embeddings), O(fi.linear_embed.embeddings)
    sparse = graph.sparse_embed([emb[offs[f]:offs[f] + vocab[f]] for f in range(len(vocab))], [idx[:, f:f + 1].cpu() for f in range(len(vocab))])
    linear = sum(lin[loff[f]:loff[f] + vocab[f]][idx[:, f].cpu()] for f in range(len(vocab)))
    cin_out = graph.cin(torch.cat(sparse, 1), [O(w)[0] for w in b.cin.conv_kernels], [O(v) for v in b.cin.conv_biases],
                        O(b.cin.logit_kernel), O(b.cin.logit_bias))
    x = graph.stack_layer([D64(dense)[:, i:i + 1] for i in range(3)] + sparse)
    for h in b.dnn.hidden_list:
        yy = x @ O(h.dense.kernel) + O(h.dense.bias)
        x = torch.relu(x + yy) if x.shape == yy.shape else torch.relu(yy)
    dnn_out = x @ O(b.dnn.logit_layer.kernel) + O(b.dnn.logit_layer.bias)
    p64 = torch.sigmoid(linear + cin_out + dnn_out)[:, 0]
    y64 = y.cpu().double()
    reg64 = sum(1e-3 * emb[offs[f]:offs[f] + vocab[f]].square().sum() for f in range(len(vocab)))
    loss64 = -(y64 * torch.log(p64) + (1 - y64) * torch.log(1 - p64)).mean() + reg64
    loss64.backward()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    after64 = {}
    for n in names:
        g = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        m, v = (1 - b1) * g, (1 - b2) * g * g
        after64[n] = before[n] - lr * (m / (1 - b1)) / ((v / (1 - b2)).sqrt() + eps)
    # ---- HIP path: same step
    opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(b1, b2), eps=eps)
    opt.zero_grad()
    out = model(dense, idx)
    loss = torch.nn.functional.binary_cross_entropy(out[:, 0], y) + collect_regularization_loss(model)
    assert abs(float(loss) - float(loss64)) < 1e-5 * abs(float(loss64))
    loss.backward()
    opt.step()
    for n, p in model.named_parameters():
        got = p.detach().cpu().double()
        if before[n].abs().max() > 0:
            assert rel(got, after64[n]) < 1e-5, n
        upd, upd64 = got - before[n], after64[n] - before[n]
        if upd64.abs().max() > 0:
            assert float((upd - upd64).abs().max() / upd64.abs().max()) < 1e-3, n     # new - old, the sensitive quantity


@pytest.mark.parametrize("name", ["FM", "DeepFM", "DCN", "XDeepFM", "AutoInt"])
def test_zoo_models_train(name):
    torch.manual_seed(0)
    vocab = [9, 4, 6, 12, 5]
    B, K = 64, 8
    info = models.make_sparse_info(vocab, embed_dim=K)
    single_linear = name == "XDeepFM"
    fi = models.FeatureInput(sparseInfo=info, useLinear=name in ("FM", "DeepFM", "XDeepFM"), useAddLinear=single_linear,
                             useFlattenLinear=single_linear)
    body = {"FM": lambda: models.FM(), "DeepFM": lambda: models.DeepFM(hidden_units=[16, 8]),
            "DCN": lambda: models.DCN(hidden_units=[16, 8], cross_hidden=2),
            "XDeepFM": lambda: models.XDeepFM(conv_size=[8, 8], hidden_units=[16, 8]),
            "AutoInt": lambda: models.AutoInt(attention_dim=8, attention_head_dim=2)}[name]()
    model = models.CTRModel(fi, body).cuda()
    dense, idx = _inputs(B, 2, vocab, seed=1)
    if name in ("FM", "AutoInt"):
        dense_in = None if name == "AutoInt" else dense
    else:
        dense_in = dense
    y = torch.tensor(np.random.default_rng(2).integers(0, 2, B), dtype=torch.float32, device="cuda")
    out = model(dense_in, idx)  # builds the lazily created weights
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        out = model(dense_in, idx)
        p = out[:, 1] if out.shape[1] == 2 else out[:, 0]
        loss = torch.nn.functional.binary_cross_entropy(p.clamp(1e-6, 1 - 1e-6), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0] * 0.9, losses[::6]


def test_afm_matches_oracle_composition():
    """AFM (models.py:141-147) through InnerLayer pairs (HIP) + AttentionBaseLayer, against the oracle graph."""
    vocab = [7, 11, 5, 13]
    B, K = 17, 8
    info = models.make_sparse_info(vocab, embed_dim=K)
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useFlattenLinear=True)
    model = models.CTRModel(fi, models.AFM()).cuda()
    _, idx = _inputs(B, 1, vocab, seed=3)
    out = model(None, idx)
    assert out.shape == (B, 1)
    D = lambda t: t.detach().cpu().double()
    emb, offs = D(fi.sparse_embed.embeddings), fi.sparse_embed.offsets.cpu()
    sparse_embed = graph.sparse_embed([emb[offs[f]:offs[f] + vocab[f]] for f in range(len(vocab))],
                                      [idx[:, f:f + 1].cpu() for f in range(len(vocab))])
    lin, loff = D(fi.linear_embed.embeddings), fi.linear_embed.offsets.cpu()
    linear = [lin[loff[f]:loff[f] + vocab[f]][idx[:, f].cpu()] for f in range(len(vocab))]  # F x [B,1]
    a = model.body.atten
    atten = graph.attention_base_layer(graph.inner_layer(sparse_embed), D(a.kernel_w), D(a.kernel_b), D(a.single_mlp_kernel),
                                       D(a.output_layer.kernel), D(a.output_layer.bias))
    want = torch.sigmoid(sum(linear) + atten)
    assert rel(out, want) < 1e-5
    # the reference's softmax runs over a size-1 axis: the score weights cannot receive gradient
    out.sum().backward()
    assert float(a.kernel_w.grad.abs().max()) == 0.0 and float(fi.sparse_embed.embeddings.grad.abs().max()) > 0.0


@pytest.mark.parametrize("name", ["NFM", "AFM", "PNN", "Wide_Deep", "DeepCross"])
def test_more_zoo_models_train(name):
    torch.manual_seed(0)
    vocab = [9, 4, 6, 12, 5]
    B, K = 64, 8
    info = models.make_sparse_info(vocab, embed_dim=K)
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useFlattenLinear=True)
    body = {"NFM": lambda: models.NFM(hidden_units=[16, 8]), "AFM": lambda: models.AFM(),
            "PNN": lambda: models.PNN(hidden_units=[16, 8], use_outer=False),
            "Wide_Deep": lambda: models.Wide_Deep(hidden_units=[16, 8]), "DeepCross": lambda: models.DeepCross()}[name]()
    model = models.CTRModel(fi, body).cuda()
    dense, idx = _inputs(B, 2, vocab, seed=1)
    y = torch.tensor(np.random.default_rng(2).integers(0, 2, B), dtype=torch.float32, device="cuda")
    out = model(dense, idx)
    opt = torch.optim.Adam(model.parameters(), lr=0.002 if name == "DeepCross" else 0.05)  # DeepCross: fixed 3x256 MLP
    losses = []
    for _ in range(25):
        opt.zero_grad()
        out = model(dense, idx)
        p = out[:, 1] if out.shape[1] == 2 else out[:, 0]
        loss = torch.nn.functional.binary_cross_entropy(p.clamp(1e-6, 1 - 1e-6), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0] * 0.9, losses[::6]


def test_reference_quirks_of_the_zoo():
    vocab = [5, 6, 7]
    info = models.make_sparse_info(vocab, embed_dim=4)
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useFlattenLinear=True).cuda()
    dense, idx = _inputs(8, 1, vocab)
    fea = fi((dense, idx))
    with pytest.raises(AttributeError):   # PNN's default use_outer=True reaches InnerLayer(use_inner=False) (models.py:50)
        models.PNN(hidden_units=[8])(fea)
    assert models.DeepCross(hidden_units=[8])(fea) is None   # models.py:57-66: the body sits under `if hidden_units is None`


def test_deepfm_config2_bf16():
    """BASELINE config 2: DeepFM (FM + 2-layer MLP), 39 fields, K=16, B=4096, bf16.  The FM layer runs its bf16-storage
    kernel (fp32 accumulate) on bf16 embeddings, the MLP runs under bf16 autocast; compared with the fp32 run of the
    same model (tolerance of the labelled bf16 mode: 2e-2 on the click probabilities)."""
    torch.manual_seed(0)
    rng = np.random.default_rng(7)
    vocab = [int(v) for v in rng.integers(10, 1000, 39)]
    B, K = 4096, 16
    fi = models.FeatureInput(sparseInfo=models.make_sparse_info(vocab, embed_dim=K), useLinear=True)
    model = models.CTRModel(fi, models.DeepFM(hidden_units=[256, 128])).cuda()
    dense, idx = _inputs(B, 13, vocab, seed=5)
    out32 = model(dense, idx)

    class Bf16Body(torch.nn.Module):
        def __init__(self, body):
            super().__init__()
            self.body = body

        def forward(self, fea):
            fea.sparse_embed = [e.bfloat16() for e in fea.sparse_embed]
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                return self.body(fea).float()

    out16 = models.CTRModel(fi, Bf16Body(model.body))(dense, idx)
    assert out16.shape == (B, 2) and torch.isfinite(out16).all()
    assert float((out16 - out32).detach().abs().max()) < 2e-2
    assert float((out16 - out32).detach().abs().max()) > 0.0  # really a different arithmetic
    # against the ORACLE (not the model itself): the reference graph in float64 on the bf16-rounded embeddings -- fp32 run at
    # 1e-5 of the unrounded oracle, bf16 run within the labelled mode's 2e-2 on the click probabilities
    b = model.body
    for rounded, got, tol in ((False, out32, 1e-5), (True, out16, 2e-2)):
        sparse, linear, dense_l = _oracle_features(fi, vocab, dense, idx, round_bf16=rounded)
        want = _oracle_merge_score(b.score, [graph.fm_layer(sparse, linear), _oracle_dnn(b.dnn, graph.stack_layer(dense_l + sparse))])
        assert float((got.detach().cpu().double() - want).abs().max()) < tol, rounded
    y = torch.tensor(rng.integers(0, 2, B), dtype=torch.float32, device="cuda")
    torch.nn.functional.binary_cross_entropy(out16[:, 1].clamp(1e-6, 1 - 1e-6), y).backward()
    g = fi.sparse_embed.embeddings.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


@pytest.mark.parametrize("extra", [[], ["--no-graph"]])
def test_train_example_runs_graph_captured_and_eager(extra):
    """examples/train_ctr.py (the counterpart of the reference's un_seq.py): by default the whole step -- embedding gather, FM /
    CIN / MLP, BCE, backward incl. the deterministic table gradients, Adam -- is captured once into a HIP graph and replayed
    (the C ABI neither allocates nor synchronises); --no-graph runs the same step eagerly.  Both must learn the synthetic
    teacher (AUC well above chance after 40 steps)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "train_ctr.py"), "--model", "XDeepFM", "--steps", "40", "--batch", "2048"]
                       + extra, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    aucs = [float(m.group(1)) for m in re.finditer(r"auc ([0-9.]+)", r.stdout)]
    assert len(aucs) >= 2 and aucs[-1] > 0.7 and aucs[-1] > aucs[0], r.stdout
