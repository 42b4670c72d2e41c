"""Committed golden vectors (tests/golden/*.npz, generator: tests/golden/make_golden.py).

CPU: both oracle restatements reproduce the fixtures; the field-index golden comes from real sklearn.
GPU: the HIP path reproduces the fixtures (norm-relative 1e-5; field-index work bit-exact)."""
import os

import numpy as np
import pytest
import torch

from oracle import closed

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda n: dict(np.load(os.path.join(GOLD, n)))


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# ------------------------------------------------------------------ CPU: oracle vs fixtures
@pytest.mark.parametrize("name", ["fm_c1_small.npz", "fm_tiny.npz"])
def test_oracle_fm(name):
    g = load(name)
    assert rel(closed.fm_fwd(g["emb"], g["lin"]), g["out"][:, 0]) < 1e-6
    demb, dlin = closed.fm_bwd(g["emb"], g["g"])
    assert rel(demb, g["demb"]) < 1e-6 and rel(dlin, g["dlin"]) < 1e-6


@pytest.mark.parametrize("name", ["dcn_small.npz", "dcn_tiny.npz"])
def test_oracle_dcn(name):
    g = load(name)
    y, _ = closed.dcn_fwd(g["x"], g["w"], g["b"])
    assert rel(y, g["y"][..., 0]) < 1e-6
    dx, dw, db = closed.dcn_bwd(g["x"], g["w"], g["b"], g["g"])
    assert rel(dx, g["dx"]) < 1e-6 and rel(dw, g["dw"]) < 1e-6 and rel(db, g["db"]) < 1e-6


def _cin_parts(g):
    L = len(g["conv"])
    return [g["W%d" % l] for l in range(L)], [g["b%d" % l] for l in range(L)], L


@pytest.mark.parametrize("name", ["cin_c4_narrow.npz", "cin_tiny.npz"])
def test_oracle_cin(name):
    g = load(name)
    Ws, bs, L = _cin_parts(g)
    assert rel(closed.cin_fwd(g["x"], Ws, bs, g["dense_w"], g["dense_b"]), g["out"]) < 1e-6
    dx, dWs, dbs, ddw, ddb = closed.cin_bwd(g["x"], Ws, bs, g["dense_w"], g["g"])
    assert rel(dx, g["dx"]) < 1e-6 and rel(ddw, g["ddense_w"]) < 1e-6 and rel(ddb, g["ddense_b"]) < 1e-6
    for l in range(L):
        assert rel(dWs[l], g["dW%d" % l]) < 1e-6 and rel(dbs[l], g["db%d" % l]) < 1e-6


@pytest.mark.parametrize("name", ["attn_c5_small.npz", "attn_default.npz"])
def test_oracle_attn(name):
    g = load(name)
    y = closed.attn_fwd(g["x"], g["Wq"], g["Wk"], g["Wr"], g["gamma"], g["beta"])
    assert rel(y, g["y"]) < 1e-6
    H, B, F, A = y.shape
    assert np.array_equal(g["flat"][:, F * A:2 * F * A], g["y"][1].reshape(B, F * A))  # head-major flatten, models.py:162
    grads = closed.attn_bwd(g["x"], g["Wq"], g["Wk"], g["Wr"], g["gamma"], g["beta"], g["dy"])
    for got, n in zip(grads, ["dx", "dWq", "dWk", "dWr", "dgamma", "dbeta"]):
        assert rel(got, g[n]) < 1e-5, n  # float32-stored fixture of a kinked (ReLU) function


def test_oracle_attn_stack():
    g = load("attn_stack_c5_small.npz")
    L = int(g["L"])
    layers = [tuple(g["%s%d" % (n, l)] for n in ["Wq", "Wk", "Wr", "gamma", "beta"]) for l in range(L)]
    assert L == 3 and layers[1][0].shape == (64, 4, 16)          # BASELINE config 5: 3 layers, 4 heads, A=16
    assert rel(closed.attn_stack_fwd(g["x"], layers), g["y"]) < 1e-6
    dx, grads = closed.attn_stack_bwd(g["x"], layers, g["dy"])
    assert rel(dx, g["dx"]) < 1e-5
    for l in range(L):
        for got, n in zip(grads[l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            assert rel(got, g["%s%d" % (n, l)]) < 1e-5, (l, n)


def test_label_encode_golden_from_sklearn():
    g = load("label_encode.npz")
    for f, n in enumerate(["C1", "C2", "C3"]):
        idx, classes = closed.label_encode(list(g["col_" + n]))
        assert np.array_equal(idx, g["idx"][:, f]), n       # bit-exact index parity with sklearn.LabelEncoder
        assert len(classes) == int(g["vocab"][f])
    offs = np.concatenate([[0], np.cumsum(g["vocab"])[:-1]])
    tables = [g["table"][o:o + v] for o, v in zip(offs, g["vocab"])]
    assert np.array_equal(closed.embed_gather(tables, g["idx"]), g["gathered"])


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fm_c1_small.npz", "fm_tiny.npz"])
def test_gpu_fm(name):
    from ml_function_amd.layers import FmLayer
    g = load(name)
    F = g["emb"].shape[1]
    emb, lin = dev(g["emb"]).requires_grad_(), dev(g["lin"]).requires_grad_()
    out = FmLayer()([[emb[:, f:f + 1, :] for f in range(F)], [lin[:, f:f + 1, None] for f in range(F)]])
    assert out.shape == g["out"].shape and rel(out, g["out"]) < 1e-5
    out.backward(dev(g["g"])[:, None, :])
    assert rel(emb.grad, g["demb"]) < 1e-5 and rel(lin.grad, g["dlin"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dcn_small.npz", "dcn_tiny.npz"])
def test_gpu_dcn(name):
    from ml_function_amd.layers import CrossLayer
    g = load(name)
    L, D = g["w"].shape
    layer = CrossLayer(cross_hidden=L)
    x = dev(g["x"]).requires_grad_()
    layer.build((None, D))
    layer.to("cuda")
    with torch.no_grad():
        for l in range(L):
            getattr(layer, "outer_weight_%d" % l).copy_(dev(g["w"][l])[:, None])
            getattr(layer, "outer_bias_%d" % l).copy_(dev(g["b"][l])[:, None])
    y = layer(x)
    assert y.shape == g["y"].shape and rel(y, g["y"]) < 1e-5
    y.backward(dev(g["g"])[..., None])
    assert rel(x.grad, g["dx"]) < 1e-5
    for l in range(L):
        assert rel(getattr(layer, "outer_weight_%d" % l).grad[:, 0], g["dw"][l]) < 1e-5
        assert rel(getattr(layer, "outer_bias_%d" % l).grad[:, 0], g["db"][l]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cin_c4_narrow.npz", "cin_tiny.npz"])
def test_gpu_cin(name):
    from ml_function_amd.layers import CIN
    g = load(name)
    Ws, bs, L = _cin_parts(g)
    layer = CIN(conv_size=[int(h) for h in g["conv"]], output_dim=1)
    x = dev(g["x"]).requires_grad_()
    layer.build(tuple(g["x"].shape))
    layer.to("cuda")
    with torch.no_grad():
        for l in range(L):
            getattr(layer, "hidden_conv_%d_kernel" % l).copy_(dev(Ws[l])[None])
            getattr(layer, "hidden_conv_%d_bias" % l).copy_(dev(bs[l]))
        layer.logit_layer_kernel.copy_(dev(g["dense_w"]))
        layer.logit_layer_bias.copy_(dev(g["dense_b"]))
    out = layer(x)
    assert out.shape == g["out"].shape and rel(out, g["out"]) < 1e-5
    out.backward(dev(g["g"]))
    assert rel(x.grad, g["dx"]) < 1e-5
    for l in range(L):
        assert rel(getattr(layer, "hidden_conv_%d_kernel" % l).grad[0], g["dW%d" % l]) < 1e-5
        assert rel(getattr(layer, "hidden_conv_%d_bias" % l).grad, g["db%d" % l]) < 1e-5
    assert rel(layer.logit_layer_kernel.grad, g["ddense_w"]) < 1e-5 and rel(layer.logit_layer_bias.grad, g["ddense_b"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["attn_c5_small.npz", "attn_default.npz"])
def test_gpu_autoint(name):
    from ml_function_amd.layers import DnnLayer, MultHeadAttentionLayer, StackLayer
    g = load(name)
    K, H, A = g["Wq"].shape
    att = MultHeadAttentionLayer(attention_dim=A, attention_head_dim=H, use_ln=True, atten_mask_mod=1)
    dnn = DnnLayer(res_unit=1, other_dense=[att])  # models.py:160-161
    x = dev(g["x"]).requires_grad_()
    att._build_device = x.device
    att.build(tuple(g["x"].shape))
    att.built = True
    with torch.no_grad():
        att.query_w.copy_(dev(g["Wq"])); att.key_w.copy_(dev(g["Wk"])); att.res_w.copy_(dev(g["Wr"]))
        att.ln_gamma.copy_(dev(g["gamma"])); att.ln_beta.copy_(dev(g["beta"]))
    y = dnn(x)
    assert y.shape == g["y"].shape and rel(y, g["y"]) < 1e-5
    flat = StackLayer(use_flat=True, axis=-1)([h.squeeze(0) for h in torch.split(y, 1, dim=0)])  # models.py:162
    assert rel(flat, g["flat"]) < 1e-5
    y.backward(dev(g["dy"]))
    for p, n in [(x, "dx"), (att.query_w, "dWq"), (att.key_w, "dWk"), (att.res_w, "dWr"), (att.ln_gamma, "dgamma"),
                 (att.ln_beta, "dbeta")]:
        assert rel(p.grad, g[n]) < 2e-5, n
    assert att.value_w.grad is None  # value_w never participates (reference behavior_layer.py:360)
    # stand-alone layer call returns [atten_v, res] like the reference
    av, res = att(x.detach())
    assert av.shape == g["y"].shape and res.shape == g["y"].shape
    assert rel(torch.relu(av + res), g["y"]) < 1e-5


def _run_stack_fixture(name, precision):
    from ml_function_amd import functional as Fn
    g = load(name)
    L = int(g["L"])
    x = dev(g["x"]).requires_grad_()
    layers = [tuple(dev(g["%s%d" % (n, l)]).requires_grad_() for n in ["Wq", "Wk", "Wr", "gamma", "beta"]) for l in range(L)]
    y = Fn.autoint_stack(x, layers, precision=precision)
    y.backward(dev(g["dy"]))
    return g, L, x, layers, y


@pytest.mark.gpu
def test_gpu_autoint_stack():
    """BASELINE config 5 as written: 3 stacked interacting layers, 4 heads, F=200, K=16, A=16; layers 2 and 3 read the
    head-major output of the layer below in place (no head-concat copy).  fp32 mode at the 1e-5 bar on the output (5e-5
    on gradients: three ReLU/LN kinks deep).  dWq of the upper layers is ill-conditioned on this fixture (its terms cancel
    to ~1e-3 of their size: the oracle graph itself evaluated in fp32 by torch-CPU is off by 1.6e-4 on dWq of layer 3), bar x10."""
    g, L, x, layers, y = _run_stack_fixture("attn_stack_c5_small.npz", "f32")
    assert y.shape == g["y"].shape and rel(y, g["y"]) < 1e-5
    assert rel(x.grad, g["dx"]) < 5e-5
    for l in range(L):
        for p, n in zip(layers[l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            assert rel(p.grad, g["%s%d" % (n, l)]) < (5e-4 if (n == "dWq" and l > 0) else 5e-5), (l, n)


@pytest.mark.gpu
def test_gpu_autoint_stack_f16_kinked_smoke():
    """The same fixture in the labelled f16-MFMA mode: outputs at 5e-3.  Its GRADIENTS are a smoke check only (finite, within
    0.25): outputs that sit within ~1e-3 of the ReLU kink land on the other side of zero in reduced precision and flip their
    whole upstream gradient -- a property of the inputs, not of the kernels.  The bar that bites is the kink-free fixture below."""
    g, L, x, layers, y = _run_stack_fixture("attn_stack_c5_small.npz", "f16_mfma")
    assert rel(y, g["y"]) < 5e-3
    assert torch.isfinite(x.grad).all() and rel(x.grad, g["dx"]) < 0.25
    for l in range(L):
        for p, n in zip(layers[l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            assert torch.isfinite(p.grad).all(), (l, n)
            if not (n == "dWq" and l > 0):      # (ill-conditioned on this fixture, see test_gpu_autoint_stack)
                assert rel(p.grad, g["%s%d" % (n, l)]) < 0.25, (l, n)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_y,tol_g", [("f32", 1e-5, 5e-5), ("f16_mfma", 5e-3, 2e-2)])
def test_gpu_autoint_stack_kink_free(precision, tol_y, tol_g):
    """Config 5's 3-layer stack on the kink-free fixture (synth.attn_stack_case(beta_shift=4, center_upper=True): every
    pre-activation >= 0.6, sigmoid scores unsaturated, all weight gradients O(0.3..10)): EVERY gradient of EVERY layer
    is held to the mode's bar -- fp32 1e-5 / 5e-5, f16-MFMA 5e-3 / 2e-2.  dWq of the layers above the
    first is ill-conditioned in any precision (tests/test_gpu_parity.py::test_attn_at_the_benchmark_shape; measured on this
    fixture: fp32 5e-5 against 3e-7 for the other gradients, f16 4.7e-2 against 1e-3): bar x5 there."""
    g, L, x, layers, y = _run_stack_fixture("attn_stack_c5_nokink.npz", precision)
    assert float(g["y"].min()) > 0.5
    assert rel(y, g["y"]) < tol_y
    assert rel(x.grad, g["dx"]) < tol_g
    for l in range(L):
        for p, n in zip(layers[l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            assert rel(p.grad, g["%s%d" % (n, l)]) < (5 * tol_g if (n == "dWq" and l > 0) else tol_g), (l, n)


@pytest.mark.gpu
def test_gpu_sparse_embed_bit_exact():
    from collections import namedtuple
    from ml_function_amd.layers import SparseEmbed
    g = load("label_encode.npz")
    Info = namedtuple("sparseFea", ["fea_name", "word_size", "cross_unit", "linear_unit"])
    infos = [Info("C%d" % (i + 1), int(v), 8, 1) for i, v in enumerate(g["vocab"])]
    emb = SparseEmbed(infos, use_flatten=False)
    idx = torch.tensor(g["idx"], device="cuda")
    emb._build_device = idx.device
    nf = len(g["vocab"])
    emb.build([(64, 1)] * nf)
    emb.built = True
    with torch.no_grad():
        emb.embeddings.copy_(dev(g["table"]))
    outs = emb([idx[:, f:f + 1] for f in range(nf)])
    assert len(outs) == nf and outs[0].shape == (64, 1, 8)
    assert np.array_equal(torch.cat(outs, 1).detach().cpu().numpy(), g["gathered"])
