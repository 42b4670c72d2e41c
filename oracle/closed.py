"""Closed-form NumPy fp64 formulas + hand-derived backward (TEST INFRASTRUCTURE).

PARITY UNPINNED -- see oracle/__init__.py.  These are the formulas of SURVEY.md
Appendix A; they are an independent second restatement of oracle/graph.py (which
follows the reference op by op) and they spell out, in NumPy, exactly the backward
recurrences the HIP kernels implement.  tests/test_oracle_*.py check
closed == graph (forward) and closed backward == torch autograd of graph (fp64).

Reference lines (relative to /root/reference/kon/model/ctr_model/layer/):
  FM   interactive_layer/interactive_layer.py:59-66,161-170
  DCN  interactive_layer/interactive_layer.py:275-282
  CIN  interactive_layer/interactive_layer.py:310-327
  ATT  behavior_layer/behavior_layer.py:292-311,356-377 + core_layer/core_layer.py:201-226
"""
import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------- FM
def fm_fwd(emb, lin=None):
    """emb [B,F,K], lin [B,F] or None -> out [B,K] = sum_{i<j} e_i*e_j + sum_f lin_f (broadcast over K)."""
    emb = np.asarray(emb, F64)
    s = emb.sum(1)
    out = 0.5 * (s * s - (emb * emb).sum(1))
    if lin is not None:
        out = out + np.asarray(lin, F64).sum(1, keepdims=True)
    return out


def fm_bwd(emb, g):
    """g [B,K] -> (demb [B,F,K] = g*(S - e_f), dlin [B,F] = sum_k g)."""
    emb = np.asarray(emb, F64)
    g = np.asarray(g, F64)
    s = emb.sum(1, keepdims=True)
    demb = g[:, None, :] * (s - emb)
    dlin = np.repeat(g.sum(1, keepdims=True), emb.shape[1], axis=1)
    return demb, dlin


def fm_pairs_fwd(emb):
    """InnerLayer(use_add=False): [B, C(F,2), K] pair products in combinations order (i asc, then j asc)."""
    emb = np.asarray(emb, F64)
    F = emb.shape[1]
    iu, ju = np.triu_indices(F, 1)
    return emb[:, iu, :] * emb[:, ju, :]


def fm_pairs_bwd(emb, gp):
    """gp [B,P,K] -> demb [B,F,K]."""
    emb = np.asarray(emb, F64)
    gp = np.asarray(gp, F64)
    F = emb.shape[1]
    iu, ju = np.triu_indices(F, 1)
    demb = np.zeros_like(emb)
    np.add.at(demb, (slice(None), iu), gp * emb[:, ju, :])
    np.add.at(demb, (slice(None), ju), gp * emb[:, iu, :])
    return demb


# --------------------------------------------------------------------------- DCN
def dcn_fwd(x, w, b):
    """x [B,D], w,b [L,D] -> (y [B,D], s [B,L]) with x_{l+1} = x0*(x_l.w_l) + x_l + b_l."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    b = np.asarray(b, F64)
    xl = x
    s = np.zeros((x.shape[0], w.shape[0]), F64)
    for l in range(w.shape[0]):
        s[:, l] = xl @ w[l]
        xl = x * s[:, l:l + 1] + xl + b[l]
    return xl, s


def dcn_bwd(x, w, b, g):
    """g [B,D] = dL/dy -> (dx [B,D], dw [L,D], db [L,D])."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    b = np.asarray(b, F64)
    g = np.asarray(g, F64)
    L = w.shape[0]
    xs = [x]
    ss = []
    for l in range(L):
        sl = xs[-1] @ w[l]
        ss.append(sl)
        xs.append(x * sl[:, None] + xs[-1] + b[l])
    gx = g.copy()
    dx0 = np.zeros_like(x)
    dw = np.zeros_like(w)
    db = np.zeros_like(b)
    for l in range(L - 1, -1, -1):
        db[l] = gx.sum(0)
        ds = (gx * x).sum(1)  # [B]
        dx0 += gx * ss[l][:, None]
        dw[l] = (xs[l] * ds[:, None]).sum(0)
        gx = gx + ds[:, None] * w[l][None, :]
    return dx0 + gx, dw, db


# --------------------------------------------------------------------------- CIN
def cin_fwd(x, Ws, bs, dense_w=None, dense_b=None, output_dim=1, return_maps=False):
    """x [B,F,K]; Ws[l] [Hp*F, H] (c = h*F+f); bs[l] [H]; dense_w [L*K,1]; dense_b [1].

    x^l[b,n,k] = sum_{h,f} W_l[h*F+f, n] x^{l-1}[b,h,k] x[b,f,k] + bias_l[n];  p_l[b,k] = sum_n x^l[b,n,k].
    """
    x = np.asarray(x, F64)
    B, F, K = x.shape
    pre = x
    maps, pools = [], []
    for W, bias in zip(Ws, bs):
        W = np.asarray(W, F64)
        Hp = pre.shape[1]
        W3 = W.reshape(Hp, F, -1)
        nxt = np.einsum('bhk,bfk,hfn->bnk', pre, x, W3, optimize=True) + np.asarray(bias, F64)[None, :, None]
        maps.append(nxt)
        pools.append(nxt.sum(1))
        pre = nxt
    P = np.concatenate(pools, axis=-1)
    out = P
    if output_dim == 1:
        out = P @ np.asarray(dense_w, F64) + np.asarray(dense_b, F64)
    if return_maps:
        return out, maps, P
    return out


def cin_bwd(x, Ws, bs, dense_w, g, output_dim=1):
    """g: [B,1] (output_dim==1) or [B,L*K].  Returns (dx, dWs, dbs, ddense_w, ddense_b).

    G^l = dP_l (broadcast over n) + Gx^l;  dbias_l = sum_{b,k} G^l;  dW_l[c,n] = sum_{b,k} Z_l[b,k,c] G^l[b,n,k];
    dZ_l[b,k,c] = sum_n G^l[b,n,k] W_l[c,n];  Gx^{l-1}[b,h,k] = sum_f dZ_l[b,k,hF+f] x[b,f,k];
    dX[b,f,k] += sum_h dZ_l[b,k,hF+f] x^{l-1}[b,h,k]  (layer 1: Gx^0 is added to dX too).
    """
    x = np.asarray(x, F64)
    g = np.asarray(g, F64)
    B, F, K = x.shape
    L = len(Ws)
    _, maps, P = cin_fwd(x, Ws, bs, dense_w, np.zeros(1), output_dim, return_maps=True)
    if output_dim == 1:
        dw_d = P.T @ g  # [L*K,1]
        db_d = g.sum(0)
        dP = g @ np.asarray(dense_w, F64).T  # [B, L*K]
    else:
        dw_d = db_d = None
        dP = g
    dx = np.zeros_like(x)
    dWs = [None] * L
    dbs = [None] * L
    Gx = None
    for l in range(L - 1, -1, -1):
        W = np.asarray(Ws[l], F64)
        H = W.shape[1]
        pre = x if l == 0 else maps[l - 1]
        Hp = pre.shape[1]
        G = np.repeat(dP[:, None, l * K:(l + 1) * K], H, axis=1)  # [B,H,K]
        if Gx is not None:
            G = G + Gx
        dbs[l] = G.sum((0, 2))
        W3 = W.reshape(Hp, F, H)
        dWs[l] = np.einsum('bhk,bfk,bnk->hfn', pre, x, G, optimize=True).reshape(Hp * F, H)
        dZ = np.einsum('bnk,hfn->bhfk', G, W3, optimize=True)  # [B,Hp,F,K]
        Gx = np.einsum('bhfk,bfk->bhk', dZ, x, optimize=True)
        dx += np.einsum('bhfk,bhk->bfk', dZ, pre, optimize=True)
    dx += Gx  # layer 1: x^{0} is x itself
    return dx, dWs, dbs, dw_d, db_d


# --------------------------------------------------------------------------- AutoInt interacting layer
def attn_fwd(x, Wq, Wk, Wr, gamma, beta, use_scale=True, use_res=True, use_ln=True, eps=1e-3, return_saved=False):
    """x [B,F,K]; Wq/Wk/Wr [K,H,A]; gamma,beta [A] -> y [H,B,F,A] = relu(res + LN(sigmoid(q k^T/sqrt(A)) v)), v == k."""
    x = np.asarray(x, F64)
    Wq, Wk = np.asarray(Wq, F64), np.asarray(Wk, F64)
    A = Wq.shape[-1]
    q = np.einsum('bfk,kma->mbfa', x, Wq)
    kk = np.einsum('bfk,kma->mbfa', x, Wk)
    scale = 1.0 / np.sqrt(A) if use_scale else 1.0
    pre = np.einsum('mbfa,mbga->mbfg', q, kk) * scale
    S = 1.0 / (1.0 + np.exp(-pre))
    av = np.einsum('mbfg,mbga->mbfa', S, kk)
    if use_ln:
        mu = av.mean(-1, keepdims=True)
        var = ((av - mu) ** 2).mean(-1, keepdims=True)
        rstd = 1.0 / np.sqrt(var + eps)
        xhat = (av - mu) * rstd
        ln = xhat * np.asarray(gamma, F64) + np.asarray(beta, F64)
    else:
        rstd = xhat = None
        ln = av
    if use_res:
        r = np.einsum('bfk,kma->mbfa', x, np.asarray(Wr, F64))
        z = r + ln
    else:
        z = ln
    y = np.maximum(z, 0.0)
    if return_saved:
        return y, dict(q=q, kk=kk, S=S, av=av, rstd=rstd, xhat=xhat, z=z, scale=scale)
    return y


def attn_bwd(x, Wq, Wk, Wr, gamma, beta, dy, use_scale=True, use_res=True, use_ln=True, eps=1e-3):
    """dy [H,B,F,A] -> (dx, dWq, dWk, dWr, dgamma, dbeta).  K and V share key_w, so their grads add."""
    x = np.asarray(x, F64)
    Wq, Wk = np.asarray(Wq, F64), np.asarray(Wk, F64)
    dy = np.asarray(dy, F64)
    y, sv = attn_fwd(x, Wq, Wk, Wr, gamma, beta, use_scale, use_res, use_ln, eps, return_saved=True)
    q, kk, S = sv['q'], sv['kk'], sv['S']
    dz = dy * (sv['z'] > 0)
    if use_ln:
        xhat, rstd = sv['xhat'], sv['rstd']
        dgamma = (dz * xhat).sum((0, 1, 2))
        dbeta = dz.sum((0, 1, 2))
        dxhat = dz * np.asarray(gamma, F64)
        dav = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    else:
        dgamma = dbeta = None
        dav = dz
    dS = np.einsum('mbfa,mbga->mbfg', dav, kk)
    dV = np.einsum('mbfg,mbfa->mbga', S, dav)
    dpre = dS * S * (1.0 - S) * sv['scale']
    dq = np.einsum('mbfg,mbga->mbfa', dpre, kk)
    dkk = np.einsum('mbfg,mbfa->mbga', dpre, q) + dV
    dWq = np.einsum('bfk,mbfa->kma', x, dq)
    dWk = np.einsum('bfk,mbfa->kma', x, dkk)
    dx = np.einsum('mbfa,kma->bfk', dq, Wq) + np.einsum('mbfa,kma->bfk', dkk, Wk)
    if use_res:
        Wr = np.asarray(Wr, F64)
        dWr = np.einsum('bfk,mbfa->kma', x, dz)
        dx = dx + np.einsum('mbfa,kma->bfk', dz, Wr)
    else:
        dWr = None
    return dx, dWq, dWk, dWr, dgamma, dbeta


def head_concat(y):
    """[H,B,F,A] -> [B,F,H*A] (feature h*A + a), behavior_layer.py:973."""
    H, B, F, A = y.shape
    return np.transpose(y, (1, 2, 0, 3)).reshape(B, F, H * A)


def head_split(xg, H):
    """inverse of head_concat for gradients: [B,F,H*A] -> [H,B,F,A]."""
    B, F, HA = xg.shape
    return np.transpose(xg.reshape(B, F, H, HA // H), (2, 0, 1, 3))


def attn_stack_fwd(x, layers, return_inputs=False):
    """Stack of interacting layers (extension, see oracle/graph.py:autoint_stack).  layers: list of
    (Wq, Wk, Wr, gamma, beta); returns the last [H,B,F,A] (and every layer's input when return_inputs)."""
    xs, y = [], None
    for (Wq, Wk, Wr, gamma, beta) in layers:
        xs.append(np.asarray(x, F64))
        y = attn_fwd(x, Wq, Wk, Wr, gamma, beta)
        x = head_concat(y)
    return (y, xs) if return_inputs else y


def attn_stack_bwd(x, layers, dy):
    """dy [H,B,F,A] of the last layer -> (dx, [per-layer (dWq, dWk, dWr, dgamma, dbeta)])."""
    _, xs = attn_stack_fwd(x, layers, return_inputs=True)
    grads = [None] * len(layers)
    g = np.asarray(dy, F64)
    for l in range(len(layers) - 1, -1, -1):
        Wq, Wk, Wr, gamma, beta = layers[l]
        dx, dWq, dWk, dWr, dg, db = attn_bwd(xs[l], Wq, Wk, Wr, gamma, beta, g)
        grads[l] = (dWq, dWk, dWr, dg, db)
        if l > 0:
            g = head_split(dx, np.asarray(layers[l - 1][0]).shape[1])
    return dx, grads


# --------------------------------------------------------------------------- field-index work (N1)
def label_encode(column):
    """data_prepare.py:91-93 -- fillna('-1') -> astype(str) -> sklearn LabelEncoder: rank in the
    lexicographically sorted set of distinct strings ('10' < '2').  column: sequence of str/None."""
    vals = ['-1' if v is None else str(v) for v in column]
    classes = sorted(set(vals))
    index = {v: i for i, v in enumerate(classes)}
    return np.asarray([index[v] for v in vals], dtype=np.int64), classes


def embed_gather(tables, idx):
    """tables: list of F [V_f,K] arrays; idx [B,F] int -> [B,F,K] (bit-exact copy of rows)."""
    return np.stack([np.asarray(t)[np.asarray(idx)[:, f]] for f, t in enumerate(tables)], axis=1)


def embed_scatter_add(idx, g, vocab_sizes):
    """g [B,F,K] -> list of F dense table gradients [V_f,K] (sum of rows per index, fp64)."""
    out = []
    for f, v in enumerate(vocab_sizes):
        d = np.zeros((v, g.shape[2]), F64)
        np.add.at(d, np.asarray(idx)[:, f], np.asarray(g, F64)[:, f, :])
        out.append(d)
    return out
