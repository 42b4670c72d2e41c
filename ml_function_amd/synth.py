"""Seeded synthetic inputs of the shapes named in BASELINE.json.

Embeddings ~ U(-0.05, 0.05) (the glorot limit of SparseEmbed's initializer for a ~2.4k vocabulary,
reference interactive_layer.py:215), weights glorot-uniform per shape, biases zero (Keras defaults
of Conv1D / CrossLayer, interactive_layer.py:267-271,308), upstream gradient ~ N(0,1).
Seed 2020 is the reference's seed literal (interactive_layer.py:38).
"""
import numpy as np

SEED = 2020


def glorot_uniform(rng, shape, fan_in=None, fan_out=None):
    """Keras glorot_uniform: U(-l, l), l = sqrt(6/(fan_in+fan_out)); receptive-field rule for rank>2."""
    if fan_in is None:
        if len(shape) == 2:
            fan_in, fan_out = shape
        else:
            rf = int(np.prod(shape[:-2]))
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def embeddings(rng, B, F, K, dist="uniform"):
    if dist == "normal":
        return rng.standard_normal((B, F, K)).astype(np.float32)
    return rng.uniform(-0.05, 0.05, size=(B, F, K)).astype(np.float32)


def fm_case(B, F, K, seed=SEED, dist="uniform"):
    rng = np.random.default_rng(seed)
    return dict(emb=embeddings(rng, B, F, K, dist),
                lin=rng.uniform(-0.05, 0.05, size=(B, F)).astype(np.float32),
                g=rng.standard_normal((B, K)).astype(np.float32))


def dcn_case(B, D, L, seed=SEED, dist="uniform", zero_bias=False):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, D)) if dist == "normal" else rng.uniform(-0.05, 0.05, size=(B, D))).astype(np.float32)
    w = np.stack([glorot_uniform(rng, (D, 1))[:, 0] for _ in range(L)])
    b = np.zeros((L, D), np.float32) if zero_bias else rng.uniform(-0.01, 0.01, size=(L, D)).astype(np.float32)
    return dict(x=x, w=w, b=b, g=rng.standard_normal((B, D)).astype(np.float32))


def cin_case(B, F, K, conv_size, seed=SEED, dist="uniform", zero_bias=False, output_dim=1):
    rng = np.random.default_rng(seed)
    x = embeddings(rng, B, F, K, dist)
    Ws, bs = [], []
    hp = F
    for h in conv_size:
        Ws.append(glorot_uniform(rng, (1, hp * F, h))[0])  # Keras Conv1D kernel [1, C, H]
        bs.append(np.zeros(h, np.float32) if zero_bias else rng.uniform(-0.01, 0.01, size=h).astype(np.float32))
        hp = h
    L = len(conv_size)
    dense_w = glorot_uniform(rng, (L * K, 1))
    dense_b = np.zeros(1, np.float32) if zero_bias else rng.uniform(-0.01, 0.01, size=1).astype(np.float32)
    g = rng.standard_normal((B, 1) if output_dim == 1 else (B, L * K)).astype(np.float32)
    return dict(x=x, Ws=Ws, bs=bs, dense_w=dense_w, dense_b=dense_b, g=g)


def attn_case(B, F, K, H, A, seed=SEED, dist="uniform"):
    rng = np.random.default_rng(seed)
    x = embeddings(rng, B, F, K, dist)
    mk = lambda: glorot_uniform(rng, (K, H, A), fan_in=H * K, fan_out=A * K)
    return dict(x=x, Wq=mk(), Wk=mk(), Wr=mk(),
                gamma=(1.0 + 0.1 * rng.standard_normal(A)).astype(np.float32),
                beta=(0.1 * rng.standard_normal(A)).astype(np.float32),
                dy=rng.standard_normal((H, B, F, A)).astype(np.float32))


def attn_stack_case(B, F, K, H, A, L, seed=SEED, dist="uniform", beta_shift=0.0, center_upper=False):
    """L stacked interacting layers (BASELINE config 5: L=3): layer 0 reads [B,F,K], layers l > 0 the head-concat
    [B,F,H*A] of the layer below.  Returns x, layers = [(Wq, Wk, Wr, gamma, beta)], dy [H,B,F,A].
    beta_shift > 0 moves every layer's LayerNorm offset up so that no output sits near the ReLU kink ("kink-free" inputs of
    the f16 gradient tests: an element that lands on the other side of zero in reduced precision flips its whole upstream
    gradient, which says nothing about the kernels).  center_upper=True makes the projection weights of layers l > 0 zero-sum
    over their input features, so the shifted outputs of the layer below (~beta_shift + noise) do not saturate the sigmoid
    scores: x W == (x - shift) W, the scores stay O(1) and the upper layers' weight gradients stay well conditioned."""
    rng = np.random.default_rng(seed)
    x = embeddings(rng, B, F, K, dist)
    layers, kin = [], K
    for _ in range(L):
        def mk():
            w = glorot_uniform(rng, (kin, H, A), fan_in=H * kin, fan_out=A * kin)
            if center_upper and layers:
                w = (w - w.mean(0, keepdims=True)).astype(np.float32)
            return w
        layers.append((mk(), mk(), mk(), (1.0 + 0.1 * rng.standard_normal(A)).astype(np.float32),
                       (0.1 * rng.standard_normal(A) + beta_shift).astype(np.float32)))
        kin = H * A
    return dict(x=x, layers=layers, dy=rng.standard_normal((H, B, F, A)).astype(np.float32))
