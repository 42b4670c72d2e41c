"""torch.autograd.Function wrappers over the C ABI (include/fil.h).

Every function requires CUDA tensors and the HIP library; there is no eager/CPU fallback (the oracle lives
in oracle/ and is test infrastructure only).  Tensors are made contiguous; dtypes are fp32 unless noted.
"""
import ctypes

import os

import time

import torch

from . import _lib
from ._lib import FIL_BF16, FIL_F32, FilError, check, int_array, ptr, ptr_array, stream_ptr


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FilError("ml_function_amd runs on the GPU only (got a %s tensor); there is no CPU fallback"
                           % t.device.type)


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise FilError("expected float32, got %s" % t.dtype)
    return t.contiguous()


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Scratch workspaces (NOT the `saved` buffers, which live until the backward) are kept per (device, stream, call site) and only ever
# grow: consecutive calls on one stream are ordered by the stream, so they can share the bytes, and a layer call no longer pays an
# allocator round trip per direction.  Under stream capture the cache is left alone (a block allocated there belongs to the graph's
# private pool): the call allocates as before.
_WS_CACHE = {}
_WS_LAST_USE = {}      # (device, stream) -> monotonic time of the last _scratch() call on that stream
_WS_IDLE_S = 2.0       # another stream's workspaces are dropped once that stream has not asked for scratch for this long


def _scratch(nbytes, device, tag):
    nbytes = max(int(nbytes), 256)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    cur = stream_ptr()
    key = (device, cur, tag)
    now = time.monotonic()
    _WS_LAST_USE[(device, cur)] = now
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        # a (re)allocation is also the moment to let go of what IDLE streams of this device left behind: warm-up side streams
        # (capture_step, bench.py) would otherwise keep hundreds of MB at headline shapes alive for good, invisible to
        # torch.cuda.empty_cache().  A stream that is still in use (two model replicas on two streams, a side-stream warm-up
        # followed by main-stream eager steps) keeps its workspaces: only streams that have not asked for scratch for _WS_IDLE_S
        # are evicted, so two live streams no longer free each other's hot buffers on every call.  Work still queued on an
        # evicted stream keeps its bytes: the caching allocator does not hand a block out again before the stream it was
        # allocated on has passed the free.
        for k in [k for k in _WS_CACHE if k[0] == device and k[1] != cur and now - _WS_LAST_USE.get((k[0], k[1]), 0.0) > _WS_IDLE_S]:
            del _WS_CACHE[k]
        ws = _WS_CACHE[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def release_scratch():
    """Drop every cached scratch workspace (they are re-created on demand)."""
    _WS_CACHE.clear()
    _WS_LAST_USE.clear()


# --------------------------------------------------------------------------------------------- A1  FM
class _FmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, lin):
        _require_cuda(emb, lin)
        if emb.dtype not in (torch.float32, torch.bfloat16):
            raise FilError("fm: emb must be float32 or bfloat16, got %s" % emb.dtype)
        emb = emb.contiguous()
        lin = None if lin is None else _f32c(lin.float() if lin.dtype != torch.float32 else lin)
        B, F, K = emb.shape
        if lin is not None and tuple(lin.shape) != (B, lin.shape[1]):
            raise FilError("fm: lin must be [B, n_linear]")
        out = torch.empty((B, K), dtype=emb.dtype, device=emb.device)
        dt = FIL_F32 if emb.dtype == torch.float32 else FIL_BF16
        lib = _lib.load()
        if lin is not None and lin.shape[1] != F:
            # the reference's Add takes any number of [B,1,1] linear terms; the kernel sums F columns
            lin_k = torch.zeros((B, F), dtype=torch.float32, device=emb.device)
            lin_k[:, 0] = lin.sum(1)
        else:
            lin_k = lin
        check(lib.fil_fm_fwd(ptr(emb), ptr(lin_k), ptr(out), B, F, K, dt, stream_ptr()), "fil_fm_fwd")
        ctx.save_for_backward(emb)
        ctx.lin_shape = None if lin is None else tuple(lin.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        B, F, K = emb.shape
        g = g.to(emb.dtype).contiguous()
        demb = torch.empty_like(emb)
        dlin_k = None
        if ctx.lin_shape is not None and ctx.needs_input_grad[1]:
            dlin_k = torch.empty((B, F), dtype=torch.float32, device=emb.device)
        dt = FIL_F32 if emb.dtype == torch.float32 else FIL_BF16
        check(_lib.load().fil_fm_bwd(ptr(emb), ptr(g), ptr(demb), ptr(dlin_k), B, F, K, dt, stream_ptr()), "fil_fm_bwd")
        dlin = None
        if dlin_k is not None:
            n = ctx.lin_shape[1]
            dlin = dlin_k if n == F else dlin_k[:, :1].expand(B, n).contiguous()
        return demb, dlin


def fm(emb, lin=None):
    """emb [B,F,K] (fp32/bf16), lin [B,n] fp32 or None -> [B,K]: sum_{i<j} e_i*e_j + sum of linear terms."""
    return _FmFn.apply(emb, lin)


class _FmPairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb):
        _require_cuda(emb)
        emb = _f32c(emb)
        B, F, K = emb.shape
        pairs = torch.empty((B, F * (F - 1) // 2, K), dtype=torch.float32, device=emb.device)
        check(_lib.load().fil_fm_pairs_fwd(ptr(emb), ptr(pairs), B, F, K, stream_ptr()), "fil_fm_pairs_fwd")
        ctx.save_for_backward(emb)
        return pairs

    @staticmethod
    def backward(ctx, gp):
        (emb,) = ctx.saved_tensors
        B, F, K = emb.shape
        demb = torch.empty_like(emb)
        check(_lib.load().fil_fm_pairs_bwd(ptr(emb), ptr(_f32c(gp)), ptr(demb), B, F, K, stream_ptr()), "fil_fm_pairs_bwd")
        return demb


def fm_pairs(emb):
    """emb [B,F,K] -> [B, F(F-1)/2, K] pair products in itertools.combinations order."""
    return _FmPairsFn.apply(emb)


# --------------------------------------------------------------------------------------------- A2  DCN
class _DcnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _require_cuda(x, w, b)
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        B, D = x.shape
        L = w.shape[0]
        y = torch.empty_like(x)
        s = torch.empty((B, L), dtype=torch.float32, device=x.device)
        check(_lib.load().fil_dcn_fwd(ptr(x), ptr(w), ptr(b), ptr(y), ptr(s), B, D, L, stream_ptr()), "fil_dcn_fwd")
        ctx.save_for_backward(x, w, b, s)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, b, s = ctx.saved_tensors
        B, D = x.shape
        L = w.shape[0]
        lib = _lib.load()
        g = _f32c(g)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        nws = lib.fil_dcn_bwd_workspace_bytes(B, D, L)
        ws = _scratch(nws, x.device, "dcn_bwd")
        check(lib.fil_dcn_bwd(ptr(x), ptr(w), ptr(b), ptr(s), ptr(g), ptr(dx), ptr(dw), ptr(db), B, D, L, ptr(ws), nws,
                              stream_ptr()), "fil_dcn_bwd")
        return dx, dw, db


def dcn_cross(x, w, b):
    """x [B,D], w,b [L,D] -> x_L [B,D] with x_{l+1} = x0*(x_l.w_l) + x_l + b_l."""
    return _DcnFn.apply(x, w, b)


# --------------------------------------------------------------------------------------------- A3  CIN
CIN_BF16X3 = 2          # fil.h FIL_CIN_BF16X3: the labelled split-bf16 mode of the merged quadratic tail's GEMMs (csrc/cin_qsplit.h)
CIN_X_TRANSPOSED = 16   # fil.h FIL_CIN_X_TRANSPOSED: x handed over as [B*K, F] (embed_gather(emit_xt=True))
CIN_TAIL_ALWAYS = 64    # fil.h FIL_CIN_TAIL_ALWAYS: the tails wherever they are defined, whatever the batch size (tests, smoke)
CIN_NOQTAIL = 256       # fil.h FIL_CIN_NOQTAIL: three-layer nets on the F+1-column fused tail instead of the quadratic tail
CIN_NOQMERGE = 512      # fil.h FIL_CIN_NOQMERGE: round 3's two launches per direction for the quadratic tail (no 256-column forward with the pools and the head in its epilogue, no 256-column dW, no two-pass dZ)


def cin_grad_ready_points(B, F, K, H, mode=0):
    """Where in fil_cin_bwd each grad_ready slot is recorded (fil.h): list of L+1 ordinals, [l] for layer l and [L] for the dense
    head; slots with equal ordinals become final together, so a data-parallel caller reduces them with one collective."""
    lib = _lib.load()
    L = len(H)
    pts = (ctypes.c_int * (L + 1))()
    n = lib.fil_cin_grad_ready_points(int(B), int(F), int(K), L, int_array(list(H)), int(mode), pts)
    if n < 0:
        raise FilError("fil_cin_grad_ready_points: %s" % lib.fil_last_error().decode())
    return [int(v) for v in pts]


def cin_forward_raw(x, Ws, bs, dense_w, dense_b, output_dim=1, mode=0, xt=None):
    """Raw forward through the C ABI.  Returns (out [B,1] or None, pooled [B,L*K], saved uint8 buffer).
    xt (optional): x already transposed to [B*K, F] by the gather that produced it; the kernels then read it in place."""
    lib = _lib.load()
    B, F, K = x.shape
    if xt is not None:
        if tuple(xt.shape) != (B * K, F) or xt.dtype != torch.float32 or not xt.is_contiguous() or xt.device != x.device:
            raise FilError("cin: xt must be a contiguous float32 [B*K, F] = [%d, %d] tensor on %s" % (B * K, F, x.device))
        mode |= CIN_X_TRANSPOSED
    L = len(Ws)
    H = [int(w.shape[1]) for w in Ws]
    hp = F
    for l, w in enumerate(Ws):
        if tuple(w.shape) != (hp * F, H[l]) or tuple(bs[l].shape) != (H[l],):
            raise FilError("cin: W[%d] must be [%d,%d] and bias[%d] [%d]" % (l, hp * F, H[l], l, H[l]))
        hp = H[l]
    Harr = int_array(H)
    saved = _workspace(lib.fil_cin_saved_bytes(B, F, K, L, Harr), x.device)
    nws = lib.fil_cin_fwd_workspace_bytes(B, F, K, L, Harr)
    if nws == 0 and B > 0:
        raise FilError("cin: %s" % lib.fil_last_error().decode())
    ws = _scratch(nws, x.device, "cin_fwd")
    pooled = torch.empty((B, L * K), dtype=torch.float32, device=x.device)
    out = torch.empty((B, 1), dtype=torch.float32, device=x.device) if output_dim == 1 else None
    check(lib.fil_cin_fwd(ptr(xt if xt is not None else x), ptr_array(Ws), ptr_array(bs), ptr(dense_w), ptr(dense_b), ptr(out),
                          ptr(pooled), ptr(saved), B, F, K, L, Harr, output_dim, mode, ptr(ws), nws, stream_ptr()), "fil_cin_fwd")
    return out, pooled, saved


def cin_backward_raw(x, Ws, bs, dense_w, pooled, saved, g, output_dim=1, mode=0, grads=None, ready_events=None, xt=None):
    """Raw backward.  grads (optional): dict with preallocated 'dx','dW'(list),'db'(list),'ddw','ddb' tensors
    (e.g. views into one flat all-reduce bucket).  ready_events (optional): L+1 torch.cuda.Event objects (or None
    entries), recorded as each layer's / the head's parameter gradients become final (fil.h: grad_ready_events).
    Returns the dict."""
    lib = _lib.load()
    B, F, K = x.shape
    L = len(Ws)
    H = [int(w.shape[1]) for w in Ws]
    Harr = int_array(H)
    if grads is None:
        grads = dict(dx=torch.empty_like(x), dW=[torch.empty_like(w) for w in Ws], db=[torch.empty_like(b) for b in bs],
                     ddw=torch.empty((L * K, 1), dtype=torch.float32, device=x.device) if output_dim == 1 else None,
                     ddb=torch.empty((1,), dtype=torch.float32, device=x.device) if output_dim == 1 else None)
    nws = lib.fil_cin_bwd_workspace_bytes(B, F, K, L, Harr)
    ws = _scratch(nws, x.device, "cin_bwd")
    evs = None
    if ready_events is not None:
        if len(ready_events) != L + 1:
            raise FilError("cin: ready_events must have L+1 = %d entries" % (L + 1))
        for e in ready_events:      # a torch event only owns a hipEvent_t once it has been recorded
            if e is not None and not e.cuda_event:
                e.record()
        evs = (ctypes.c_void_p * (L + 1))(*[None if e is None else e.cuda_event for e in ready_events])
    if xt is not None:
        mode |= CIN_X_TRANSPOSED
    check(lib.fil_cin_bwd(ptr(xt if xt is not None else x), ptr_array(Ws), ptr_array(bs), ptr(dense_w), ptr(pooled), ptr(saved), ptr(g),
                          ptr(grads["dx"]), ptr_array(grads["dW"]), ptr_array(grads["db"]), ptr(grads["ddw"]),
                          ptr(grads["ddb"]), B, F, K, L, Harr, output_dim, mode, evs, ptr(ws), nws, stream_ptr()), "fil_cin_bwd")
    return grads


class _CinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dense_w, dense_b, output_dim, mode, xt, *params):
        L = len(params) // 2
        Ws = [_f32c(p) for p in params[:L]]
        bs = [_f32c(p) for p in params[L:]]
        _require_cuda(x, *Ws, *bs)
        x = _f32c(x)
        dense_w = _f32c(dense_w)
        dense_b = _f32c(dense_b)
        out, pooled, saved = cin_forward_raw(x, Ws, bs, dense_w, dense_b, output_dim, mode, xt=xt)
        # xt (the gather's second output, same values as x) goes through save_for_backward too: an in-place edit between the
        # forward and the backward then trips autograd's version check instead of silently feeding stale rows to the kernels
        ctx.save_for_backward(x, dense_w, pooled, saved, *Ws, *bs, *([xt] if xt is not None else []))
        ctx.cfg = (L, output_dim, mode, xt is not None)
        return out if output_dim == 1 else pooled

    @staticmethod
    def backward(ctx, g):
        L, output_dim, mode, has_xt = ctx.cfg
        x, dense_w, pooled, saved = ctx.saved_tensors[:4]
        Ws = list(ctx.saved_tensors[4:4 + L])
        bs = list(ctx.saved_tensors[4 + L:4 + 2 * L])
        xt = ctx.saved_tensors[4 + 2 * L] if has_xt else None
        gr = cin_backward_raw(x, Ws, bs, dense_w, pooled, saved, _f32c(g), output_dim, mode, xt=xt)
        return (gr["dx"], gr["ddw"], gr["ddb"], None, None, None, *gr["dW"], *gr["db"])


def cin(x, Ws, bs, dense_w=None, dense_b=None, output_dim=1, mode=0, xt=None):
    """x [B,F,K]; Ws[l] [H_{l-1}*F, H_l]; bs[l] [H_l]; dense_w [L*K,1]; dense_b [1] -> [B,1] (or pooled [B,L*K]).
    xt (optional) = x transposed to [B*K, F] by the gather that produced x (embed_gather(emit_xt=True) attaches it to its
    result as `_fil_xt`): the kernels read it in place instead of transposing x again.  Gradients still flow to x."""
    return _CinFn.apply(x, dense_w, dense_b, output_dim, mode, xt, *Ws, *bs)


# --------------------------------------------------------------------------------------------- A4  AutoInt
class _AttnFn(torch.autograd.Function):
    """fuse_relu=True: returns y = relu(res + LN(av)); fuse_relu=False: returns (LN(av), res).
    head_major=True: x is the [H',B,F,A'] output of a previous interacting layer, read in place as its head-concat
    [B,F,H'*A'] (fil.h: x_chunk = A'); the gradient comes back in the same layout."""

    @staticmethod
    def forward(ctx, x, Wq, Wk, Wr, gamma, beta, scale, eps, fuse_relu, precision=0, head_major=False):
        _require_cuda(x, Wq, Wk, Wr, gamma, beta)
        x, Wq, Wk, Wr, gamma, beta = [_f32c(t) for t in (x, Wq, Wk, Wr, gamma, beta)]
        if head_major:
            Hp, B, F, Ap = x.shape
            K, x_chunk = Hp * Ap, Ap
        else:
            B, F, K = x.shape
            x_chunk = 0
        Kw, H, A = Wq.shape
        if Kw != K:
            raise FilError("attention: weights are [%d,H,A] but the input has %d features" % (Kw, K))
        lib = _lib.load()
        y = torch.empty((H, B, F, A), dtype=torch.float32, device=x.device)
        res = None
        if not fuse_relu and Wr is not None:
            res = torch.empty((H, B, F, A), dtype=torch.float32, device=x.device)
        # the LayerNorm input is kept for the backward's LayerNorm gradient, normalised, with the rows' 1/sigma beside it
        # (H*B*F*(A+1) floats: the backward does not derive the statistics again); FIL_ATTN_SAVE_AV=0 drops both and makes the
        # backward re-run the forward into its workspace instead
        av = rstd = None
        if gamma is not None and _SAVE_AV:
            av = torch.empty((H, B, F, A), dtype=torch.float32, device=x.device)
            rstd = torch.empty((H, B, F), dtype=torch.float32, device=x.device)
        check(lib.fil_attn_fwd(ptr(x), ptr(Wq), ptr(Wk), ptr(Wr), ptr(gamma), ptr(beta), ptr(y), ptr(res), ptr(av), ptr(rstd),
                               B, F, K, H, A,
                               float(scale), float(eps), int(bool(fuse_relu)), int(precision), x_chunk, None, 0, stream_ptr()),
              "fil_attn_fwd")
        keep_y = bool(fuse_relu) and (av is not None or gamma is None)   # the fused ReLU mask is y > 0
        ctx.save_for_backward(x, Wq, Wk, *[t for t in (Wr, gamma, beta) if t is not None],
                              *([av, rstd] if av is not None else []), *([y] if keep_y else []))
        ctx.extra = (av is not None, keep_y)
        ctx.cfg = (Wr is not None, gamma is not None, float(scale), float(eps), bool(fuse_relu), int(precision), x_chunk,
                   (B, F, K))
        if fuse_relu:
            return y
        if res is None:
            return y, None
        return y, res

    @staticmethod
    def backward(ctx, dy, dres=None):
        has_res, has_ln, scale, eps, fuse_relu, precision, x_chunk, (B, F, K) = ctx.cfg
        sv = list(ctx.saved_tensors)
        x, Wq, Wk = sv[:3]
        rest = sv[3:]
        Wr = rest.pop(0) if has_res else None
        gamma = rest.pop(0) if has_ln else None
        beta = rest.pop(0) if has_ln else None
        has_av, has_y = ctx.extra
        av_saved = rest.pop(0) if has_av else None
        rstd_saved = rest.pop(0) if has_av else None
        y_saved = rest.pop(0) if has_y else None
        _, H, A = Wq.shape
        lib = _lib.load()
        dy = _f32c(dy) if dy is not None else torch.zeros((H, B, F, A), dtype=torch.float32, device=x.device)
        dres_in = None
        if not fuse_relu and has_res:
            dres_in = _f32c(dres) if dres is not None else torch.zeros((H, B, F, A), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dWq, dWk = torch.empty_like(Wq), torch.empty_like(Wk)
        dWr = torch.empty_like(Wr) if has_res else None
        dgamma = torch.empty_like(gamma) if has_ln else None
        dbeta = torch.empty_like(beta) if has_ln else None
        have_saved = int((not has_ln or av_saved is not None) and (not fuse_relu or y_saved is not None))
        nws = lib.fil_attn_bwd_workspace_bytes(B, F, K, H, A, have_saved)
        ws = _scratch(nws, x.device, "attn_bwd")
        check(lib.fil_attn_bwd(ptr(x), ptr(Wq), ptr(Wk), ptr(Wr), ptr(gamma), ptr(beta), ptr(dy), ptr(dres_in), ptr(y_saved),
                               ptr(av_saved), ptr(rstd_saved), ptr(dx),
                               ptr(dWq), ptr(dWk), ptr(dWr), ptr(dgamma), ptr(dbeta), B, F, K, H, A, scale, eps,
                               int(fuse_relu), precision, x_chunk, ptr(ws), nws, stream_ptr()), "fil_attn_bwd")
        return dx, dWq, dWk, dWr, dgamma, dbeta, None, None, None, None, None


_SAVE_AV = os.environ.get("FIL_ATTN_SAVE_AV", "1") != "0"   # read once at import


def _attn_scale(Wq, use_scale):
    return (1.0 / (Wq.shape[-1] ** 0.5)) if use_scale else 1.0


PRECISIONS = {"f32": 0, "f16_mfma": 1}


def _precision(p):
    if p not in PRECISIONS:
        raise ValueError("precision must be one of %s, got %r" % (sorted(PRECISIONS), p))
    return PRECISIONS[p]


def autoint_interact(x, Wq, Wk, Wr=None, gamma=None, beta=None, use_scale=True, eps=1e-3, precision="f32", head_major=False):
    """x [B,F,K], W* [K,H,A] -> y [H,B,F,A] = relu(x Wr + LN(sigmoid(scale q k^T) k))  (V == K projection).
    precision "f16_mfma": matrix products on the fp16 MFMA with fp32 accumulation (include/fil.h, fil_precision).
    head_major=True: x is a previous layer's [H',B,F,A'] output, consumed as its head-concat [B,F,H'*A'] without a copy."""
    return _AttnFn.apply(x, Wq, Wk, Wr, gamma, beta, _attn_scale(Wq, use_scale), eps, True, _precision(precision),
                         bool(head_major))


def autoint_stack(x, layers, use_scale=True, eps=1e-3, precision="f32"):
    """A stack of interacting layers (BASELINE config 5: 3 layers).  layers: list of (Wq, Wk, Wr, gamma, beta); layer l > 0
    takes the head-concat [B,F,H*A] of layer l-1 (ESULayer's convention, reference behavior_layer.py:973) -- read in
    place from the head-major output.  Returns the last layer's [H,B,F,A]."""
    y = x
    for l, (Wq, Wk, Wr, gamma, beta) in enumerate(layers):
        y = autoint_interact(y, Wq, Wk, Wr, gamma, beta, use_scale=use_scale, eps=eps, precision=precision, head_major=l > 0)
    return y


def mult_head_attention(x, Wq, Wk, Wr=None, gamma=None, beta=None, use_scale=True, eps=1e-3, precision="f32"):
    """Stand-alone MultHeadAttentionLayer: returns (atten_v [H,B,F,A] = LN(sigmoid(scale q k^T) k), res = x Wr or None)."""
    return _AttnFn.apply(x, Wq, Wk, Wr, gamma, beta, _attn_scale(Wq, use_scale), eps, False, _precision(precision))


class _PAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, mask_period):
        _require_cuda(q, k, v, mask)
        q, k, v, mask = _f32c(q), _f32c(k), _f32c(v), _f32c(mask)
        N, Fq, A = q.shape
        _, Fk, Av = v.shape
        out = torch.empty((N, Fq, Av), dtype=torch.float32, device=q.device)
        check(_lib.load().fil_pattn_fwd(ptr(q), ptr(k), ptr(v), ptr(mask), ptr(out), N, Fq, Fk, A, Av, float(scale), int(mask_period),
                                        stream_ptr()), "fil_pattn_fwd")
        ctx.save_for_backward(q, k, v, *([mask] if mask is not None else []))
        ctx.cfg = (float(scale), int(mask_period), mask is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        scale, mask_period, has_mask = ctx.cfg
        sv = list(ctx.saved_tensors)
        q, k, v = sv[:3]
        mask = sv[3] if has_mask else None
        N, Fq, A = q.shape
        _, Fk, Av = v.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(_lib.load().fil_pattn_bwd(ptr(q), ptr(k), ptr(v), ptr(mask), ptr(_f32c(dout)), ptr(dq), ptr(dk), ptr(dv), N, Fq, Fk, A, Av,
                                        scale, mask_period, stream_ptr()), "fil_pattn_bwd")
        return dq, dk, dv, None, None, None


def product_attention(q, k, v, use_scale=False, mask=None, mask_mod=1):
    """ProductAttentionLayer.call([q, k, v], mask) (reference behavior_layer.py:292-311): q [..., Fq, A], k [..., Fk, A],
    v [..., Fk, Av] -> sigmoid(scale q k^T (masked)) v, leading axes flattened into the kernel's item axis.
    mask_mod 1: scores @ mask (mask [..., Fk, Fk'] broadcast over the leading axes; v must have Fk' rows) -- computed as
    q (mask^T k)^T, so only k is touched; mask_mod 2: scores + mask * (-100000), mask broadcastable to [..., Fq, Fk]."""
    lead = tuple(q.shape[:-2])
    Fq, A = q.shape[-2:]
    scale = 1.0 / (A ** 0.5) if use_scale else 1.0
    kmask, period = None, 0
    if mask is not None:
        m = mask.to(torch.float32)
        if mask_mod == 1:
            k = torch.matmul(m.transpose(-1, -2), k)            # (q k^T) M == q (M^T k)^T; autograd carries dk through
        elif mask_mod == 2:
            Fk = k.shape[-2]
            if m.dim() < 2:
                m = m.reshape((1,) * (2 - m.dim()) + tuple(m.shape))
            mlead = tuple(m.shape[:-2])
            # a mask whose leading axes are a suffix of q's (e.g. [B,Fq,Fk] against [H,B,...]) is indexed n % period in the kernel
            if len(mlead) <= len(lead) and mlead == lead[len(lead) - len(mlead):]:
                period = 1
                for s_ in mlead:
                    period *= s_
                kmask = m.expand(*mlead, Fq, Fk).reshape(max(period, 1), Fq, Fk).contiguous()
                period = max(period, 1)
            else:
                kmask = m.expand(*lead, Fq, Fk).reshape(-1, Fq, Fk).contiguous()
                period = kmask.shape[0]
    if v.shape[-2] != k.shape[-2]:
        raise FilError("product_attention: k has %d rows but v has %d" % (k.shape[-2], v.shape[-2]))
    out = _PAttnFn.apply(q.reshape(-1, Fq, A), k.expand(*lead, *k.shape[-2:]).reshape(-1, k.shape[-2], A),
                         v.reshape(-1, v.shape[-2], v.shape[-1]), kmask, scale, period)
    return out.reshape(*lead, Fq, v.shape[-1])


# --------------------------------------------------------------------------------------------- N1  embeddings
# (sorted row ids, permutation) of the most recent index tensors.  A model usually looks the SAME ids up in two tables (the
# embeddings and the linear weights): their gradients need the same sort.  An entry keeps its idx tensor alive, so an equal
# (data_ptr, _version) really is the same contents; in-place updates bump the version and miss -- and every gather (forward)
# empties the cache, so that an entry only ever serves the backward passes that follow the forwards which saw these ids
# (an update of idx that bypasses the version counter, e.g. through `.data`, cannot meet a stale entry).
_SORT_CACHE = []


class _ScoreAddSigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *parts):
        _require_cuda(*parts)
        shape = parts[0].shape
        parts = [_f32c(t) for t in parts]
        n = parts[0].numel()
        out = torch.empty(shape, dtype=torch.float32, device=parts[0].device)
        pp = [ptr(t) for t in parts] + [None] * (4 - len(parts))
        check(_lib.load().fil_score_add_sigmoid_fwd(*pp, ptr(out), n, stream_ptr()), "fil_score_add_sigmoid_fwd")
        ctx.save_for_backward(out)
        ctx.n_parts = len(parts)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _f32c(g)
        dsum = torch.empty_like(out)
        check(_lib.load().fil_score_add_sigmoid_bwd(ptr(out), ptr(g), ptr(dsum), out.numel(), stream_ptr()),
              "fil_score_add_sigmoid_bwd")
        return (dsum,) * ctx.n_parts


def score_add_sigmoid(parts):
    """sigmoid(((p0 + p1) + p2) + p3) over 1..4 fp32 tensors of ONE shape (ScoreLayer(use_add=True), core_layer.py:58-84): one launch
    forward, one backward (every part receives the same gradient tensor)."""
    parts = list(parts)
    if not 1 <= len(parts) <= 4:
        raise FilError("score_add_sigmoid takes 1..4 parts, got %d" % len(parts))
    if any(t.shape != parts[0].shape for t in parts):
        raise FilError("score_add_sigmoid: the parts must have one shape, got %s" % [tuple(t.shape) for t in parts])
    return _ScoreAddSigmoidFn.apply(*parts)


def gemm_f32(a, b, trans_a=False, trans_b=False, bias=None, relu=False):
    """op(a) @ op(b) (+ bias, ReLU) in fp32 through the library's own MFMA GEMM (fil_gemm_f32, csrc/gemm.hip): a, b contiguous 2-D CUDA
    fp32 tensors; trans_a: a is [K, M]; trans_b: b is [N, K].  No autograd (the dense layers' Functions call it in both directions)."""
    _require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[0] if trans_b else b.shape[1]
    if (b.shape[1] if trans_b else b.shape[0]) != K:
        raise FilError("gemm_f32: %s x %s (trans_a=%s, trans_b=%s)" % (tuple(a.shape), tuple(b.shape), trans_a, trans_b))
    lib = _lib.load()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    nws = lib.fil_gemm_f32_workspace_bytes(M, N, K)
    ws = _scratch(nws, a.device, "gemm_f32") if nws else None
    epi = 0 if bias is None else (2 if relu else 1)
    check(lib.fil_gemm_f32(ptr(a), ptr(b), ptr(c), ptr(_f32c(bias)) if bias is not None else None, M, N, K, a.shape[1], b.shape[1], N,
                           int(bool(trans_a)), int(bool(trans_b)), epi, ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0,
                           stream_ptr()), "fil_gemm_f32")
    return c


class _DenseReluFn(torch.autograd.Function):
    """y = relu(x @ kernel + bias) -- a hidden layer of the zoo's MLPs (DnnLayer, core_layer.py:102-118,201-226) -- as a library GEMM
    with the bias + ReLU in its epilogue forward, and backward as ONE HIP pass (ReLU mask + bias gradient, fil_relu_bias_bwd) plus the
    layer's two library GEMMs.  torch composes the same layer from 3 launches forward and 5 backward.  Under autocast the GEMMs run in
    the autocast dtype (bf16), as torch.matmul would."""

    @staticmethod
    def forward(ctx, x, kernel, bias):
        _require_cuda(x, kernel, bias)
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if cd not in (torch.float32, torch.bfloat16):
            raise FilError("dense_relu: autocast dtype %s (float32 or bfloat16)" % cd)
        with torch.autocast("cuda", enabled=False):
            xc, wc = x.to(cd).contiguous(), kernel.to(cd)
            if cd == torch.float32:      # the library's own fp32 MFMA GEMM, bias + ReLU in its epilogue (csrc/gemm.hip)
                y = gemm_f32(xc, wc, bias=bias, relu=True)
            else:                        # bf16 autocast: the framework's GEMM with the same epilogue
                y = torch._addmm_activation(bias.to(cd), xc, wc, use_gelu=False)
        ctx.save_for_backward(xc, wc, y)
        ctx.cfg = (x.dtype, kernel.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc, y = ctx.saved_tensors
        xdt, wdt, bdt = ctx.cfg
        lib = _lib.load()
        B, N = y.shape
        dy = dy.to(y.dtype).contiguous()
        dz = torch.empty_like(y)
        db = torch.empty((N,), dtype=torch.float32, device=y.device)
        ws = _scratch(lib.fil_relu_bias_bwd_workspace_bytes(B, N), y.device, "relu_bias_bwd")
        check(lib.fil_relu_bias_bwd(ptr(y), ptr(dy), ptr(dz), ptr(db), B, N, FIL_F32 if y.dtype == torch.float32 else FIL_BF16, ptr(ws),
                                    ws.numel(), stream_ptr()), "fil_relu_bias_bwd")
        with torch.autocast("cuda", enabled=False):
            if y.dtype == torch.float32:
                dx = gemm_f32(dz, wc, trans_b=True).to(xdt) if ctx.needs_input_grad[0] else None      # dz W^T
                dw = gemm_f32(xc, dz, trans_a=True).to(wdt) if ctx.needs_input_grad[1] else None      # x^T dz (split over the batch)
            else:
                dx = torch.matmul(dz, wc.t()).to(xdt) if ctx.needs_input_grad[0] else None
                dw = _batch_split_xt_dz(xc, dz).to(wdt) if ctx.needs_input_grad[1] else None
        return dx, dw, db.to(bdt)


def _batch_split_xt_dz(x, dz, splits=16):
    """x^T dz for bf16 x [B, I], dz [B, O] with the reduction over the batch cut into `splits` slices: one batched library GEMM + an
    fp32 sum of the slices in slice order.  As ONE GEMM the library runs the (I x O)-tile grid of this shape -- 8...40 workgroups for a
    4096-row batch -- at 27-31 us (MLP 637-256-128, B = 4096); cut in 16 it is 15-16 us including the sum (tools/mm_forms.py), and the
    result is fp32 without a separate cast.  Small or ragged batches take the plain product."""
    B = x.shape[0]
    if B < 1024 or B % splits != 0:
        return torch.matmul(x.t(), dz)
    parts = torch.bmm(x.view(splits, B // splits, x.shape[1]).transpose(1, 2), dz.view(splits, B // splits, dz.shape[1]))
    return parts.sum(0, dtype=torch.float32)


class _DenseFn(torch.autograd.Function):
    """y = x @ kernel + bias in fp32 through the library's own GEMM, forward and backward (dx = dy W^T, dW = x^T dy split over the batch and
    summed in order, db = the column sums).  The logit heads of the zoo's MLPs are Dense(1) / Dense(2) on a [B, 64...144] input: as library
    calls the framework runs dW = x^T dy of such a layer -- 64 x 1 outputs, reduction 4096 -- in 35 us (one 16 x 64 tile walking the batch)."""

    @staticmethod
    def forward(ctx, x, kernel, bias):
        _require_cuda(x, kernel, bias)
        xc, wc = _f32c(x), _f32c(kernel)
        ctx.save_for_backward(xc, wc)
        return gemm_f32(xc, wc, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        dy = _f32c(dy)
        dx = gemm_f32(dy, wc, trans_b=True) if ctx.needs_input_grad[0] else None
        dw = gemm_f32(xc, dy, trans_a=True) if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def dense(x, kernel, bias):
    """x @ kernel + bias, x [B, in] fp32, kernel [in, units], bias [units] (csrc/gemm.hip both ways)."""
    if x.dim() != 2 or kernel.dim() != 2 or x.shape[1] != kernel.shape[0] or tuple(bias.shape) != (kernel.shape[1],):
        raise FilError("dense: x %s, kernel %s, bias %s" % (tuple(x.shape), tuple(kernel.shape), tuple(bias.shape)))
    return _DenseFn.apply(x, kernel, bias)


def dense_relu(x, kernel, bias):
    """relu(x @ kernel + bias), x [B, in] (fp32 or bf16), kernel [in, units], bias [units]."""
    if x.dim() != 2 or kernel.dim() != 2 or x.shape[1] != kernel.shape[0] or tuple(bias.shape) != (kernel.shape[1],):
        raise FilError("dense_relu: x %s, kernel %s, bias %s" % (tuple(x.shape), tuple(kernel.shape), tuple(bias.shape)))
    return _DenseReluFn.apply(x, kernel, bias)


class _MergeSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kernel, bias, *parts):
        _require_cuda(kernel, bias, *parts)
        dts = [FIL_F32 if t.dtype == torch.float32 else FIL_BF16 for t in parts]
        parts = [t.contiguous() for t in parts]
        kernel, bias = _f32c(kernel), _f32c(bias)
        B, O = parts[0].shape[0], kernel.shape[1]
        widths = [int(t.shape[1]) for t in parts]
        out = torch.empty((B, O), dtype=torch.float32, device=kernel.device)
        check(_lib.load().fil_merge_softmax_fwd(ptr_array(parts), int_array(widths), int_array(dts), len(parts), ptr(kernel), ptr(bias),
                                                ptr(out), B, O, stream_ptr()), "fil_merge_softmax_fwd")
        ctx.save_for_backward(kernel, out, *parts)
        ctx.cfg = (widths, dts)
        return out

    @staticmethod
    def backward(ctx, g):
        kernel, out, *parts = ctx.saved_tensors
        widths, dts = ctx.cfg
        lib = _lib.load()
        B, O, D = out.shape[0], out.shape[1], sum(widths)
        g = _f32c(g)
        need = ctx.needs_input_grad[2:]
        dparts = [torch.empty_like(t) if n else None for t, n in zip(parts, need)]
        dW = torch.empty_like(kernel)
        db = torch.empty((O,), dtype=torch.float32, device=kernel.device)
        ws = _scratch(lib.fil_merge_softmax_bwd_workspace_bytes(B, D, O), kernel.device, "merge_softmax_bwd")
        dp = (ctypes.c_void_p * len(parts))(*[None if t is None else t.data_ptr() for t in dparts])
        check(lib.fil_merge_softmax_bwd(ptr_array(parts), int_array(widths), int_array(dts), len(parts), ptr(kernel), ptr(out), ptr(g), dp,
                                        ptr(dW), ptr(db), B, O, ptr(ws), ws.numel(), stream_ptr()), "fil_merge_softmax_bwd")
        return (dW, db) + tuple(dparts)


def merge_softmax(parts, kernel, bias):
    """softmax(concat(parts, -1) @ kernel + bias): MergeScoreLayer.call (core_layer.py:86-100) as one launch forward and one backward.
    parts: 1..4 tensors [B, w_i], each fp32 or bf16 (under bf16 autocast a model hands a bf16 FM output next to an fp32 MLP output);
    kernel [sum w_i, O <= 8], bias [O] fp32; returns fp32 [B, O]."""
    parts = list(parts)
    if not 1 <= len(parts) <= 4:
        raise FilError("merge_softmax takes 1..4 parts, got %d" % len(parts))
    if any(t.dtype not in (torch.float32, torch.bfloat16) or t.dim() != 2 or t.shape[0] != parts[0].shape[0] for t in parts):
        raise FilError("merge_softmax: the parts must be [B, w] float32 / bfloat16 tensors, got %s" %
                       [(tuple(t.shape), t.dtype) for t in parts])
    if kernel.dim() != 2 or kernel.shape[0] != sum(t.shape[1] for t in parts) or tuple(bias.shape) != (kernel.shape[1],):
        raise FilError("merge_softmax: kernel %s / bias %s do not fit %d concatenated columns" %
                       (tuple(kernel.shape), tuple(bias.shape), sum(t.shape[1] for t in parts)))
    return _MergeSoftmaxFn.apply(kernel, bias, *parts)


class _BceMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, y, eps):
        _require_cuda(p, y)
        if p.shape != y.shape:
            raise FilError("binary_crossentropy: p is %s, y is %s" % (tuple(p.shape), tuple(y.shape)))
        p, y = _f32c(p), _f32c(y)
        if p.numel() == 0:
            raise FilError("binary_crossentropy of an empty batch")
        loss = torch.empty((1,), dtype=torch.float32, device=p.device)
        dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        check(_lib.load().fil_bce_mean_fwd(ptr(p), ptr(y), float(eps), ptr(loss), ptr(dp), p.numel(), stream_ptr()), "fil_bce_mean_fwd")
        if dp is not None:
            ctx.save_for_backward(dp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g, None, None


def binary_crossentropy(p, y, eps=1e-7):
    """mean(-(y log(pc + eps) + (1 - y) log(1 - pc + eps))), pc = clip(p, eps, 1 - eps): tf.losses.binary_crossentropy on
    probabilities as TensorFlow 2.1 computes it (the reference compiles its CTR models with it, example/ctr_example/un_seq.py:61;
    eps = the Keras backend epsilon), as one launch that also leaves d loss / d p; deterministic."""
    return _BceMeanFn.apply(p, y, eps)


_DISJOINT_CACHE = {}


def _fields_disjoint(offsets, sizes, layout_key):
    """True when every field owns its own row range of the concatenated table: offsets ascending and offsets[f] + sizes[f] <=
    offsets[f + 1].  Only then are equal row ids adjacent after a PER-FIELD sort; two fields that share or overlap a table
    (offsets[f] == offsets[g], or no `sizes` to bound the ids) put one row into two runs, and fil_embed_run_sum STORES each run's sum
    -- one would overwrite the other.  Such layouts keep the global sort.  Checked once per layout (one small device -> host copy),
    never during a stream capture (an unchecked layout then takes the global sort)."""
    if sizes is None:
        return False
    key = layout_key if layout_key is not None else (offsets.data_ptr(), offsets._version, sizes.data_ptr(), sizes._version,
                                                     int(offsets.numel()))
    hit = _DISJOINT_CACHE.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        o, z = offsets.detach().cpu().to(torch.int64), sizes.detach().cpu().to(torch.int64)
        hit = bool(o.numel() == z.numel() and (o.numel() < 2 or bool(((o[:-1] + z[:-1]) <= o[1:]).all())) and bool((z >= 0).all()))
        if len(_DISJOINT_CACHE) > 64:
            _DISJOINT_CACHE.clear()
        _DISJOINT_CACHE[key] = hit
    return hit


def _sorted_row_ids(offsets, sizes, frozen, idx, layout_key=None, n_rows=None, per_field=False):
    """layout_key: a hashable description of (offsets, sizes, frozen) -- two tables with the same field layout (SparseEmbed
    passes its word sizes / frozen flags) share the sort even though their offset tensors are different objects."""
    lib = _lib.load()
    layout = layout_key if layout_key is not None else (
        offsets.data_ptr(), offsets._version, sizes.data_ptr() if sizes is not None else 0,
        frozen.data_ptr() if frozen is not None else 0)
    per_field = per_field and _fields_disjoint(offsets, sizes, layout_key)
    key = (idx.data_ptr(), idx._version, tuple(idx.shape), layout, torch.cuda.current_stream().cuda_stream, bool(per_field))
    for k, _, out in _SORT_CACHE:
        if k == key:
            return out
    B, F = idx.shape
    if per_field and 0 < B <= 8192 and (n_rows is None or n_rows < 2 ** 31):
        # the dense (run-sum) gradient only needs equal row ids adjacent and in a fixed order, and ids of fields with disjoint row
        # ranges (checked above) never collide: one launch sorts every field's (id, position) pairs in LDS (fil.h fil_embed_sort_fields) -- no library sort
        sorted_ids = torch.empty(B * F, dtype=torch.int64, device=idx.device)
        perm = torch.empty(B * F, dtype=torch.int64, device=idx.device)
        check(lib.fil_embed_sort_fields(ptr(offsets), ptr(sizes), ptr(frozen), ptr(idx), ptr(sorted_ids), ptr(perm), B, F,
                                        int(n_rows) if n_rows is not None else 0, stream_ptr()), "fil_embed_sort_fields")
        out = (sorted_ids, perm)
        _SORT_CACHE.insert(0, (key, (idx, offsets, sizes, frozen), out))
        del _SORT_CACHE[2:]
        return out
    row_ids = torch.empty(B * F, dtype=torch.int64, device=idx.device)
    check(lib.fil_embed_row_ids(ptr(offsets), ptr(sizes), ptr(frozen), ptr(idx), ptr(row_ids), B, F, stream_ptr()), "fil_embed_row_ids")
    # (n_rows = rows of the concatenated table: when the global row ids fit 31 bits, 32-bit keys halve the radix passes of the sort)
    # -- only with a range check: `sizes` guarantees every id in [-1, n_rows); unchecked ids could fall outside 32 bits, wrap in the
    # cast, sort out of place and merge with valid rows)
    if sizes is not None and n_rows is not None and n_rows < 2 ** 31 and row_ids.numel() > 0:
        s32, perm = torch.sort(row_ids.to(torch.int32), stable=True)
        out = (s32.to(torch.int64), perm)
    else:
        out = torch.sort(row_ids, stable=True)
    _SORT_CACHE.insert(0, (key, (idx, offsets, sizes, frozen), out))
    del _SORT_CACHE[2:]
    return out


def embed_grad_rows(offsets, sizes, idx, g, frozen=None, layout_key=None, n_rows=None):
    """Deterministic embedding-table gradient as (rows [U] int64, values [U,K]): the unique global table rows the batch
    touched (sorted) and the sum of their gradient rows, contributions added in a fixed order (no atomics).  Out-of-range
    ids and frozen fields (frozen [F] uint8) contribute nothing.  The sort is torch's (plumbing); the sums are the HIP
    segment kernel (fil.h: fil_embed_row_ids / fil_embed_segment_sum)."""
    lib = _lib.load()
    B, F = idx.shape
    K = g.shape[-1]
    g = _f32c(g)
    sorted_ids, perm = _sorted_row_ids(offsets, sizes, frozen, idx, layout_key, n_rows)
    rows, counts = torch.unique_consecutive(sorted_ids, return_counts=True)
    starts = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=g.device)
    torch.cumsum(counts, 0, out=starts[1:])
    values = torch.zeros((rows.numel(), K), dtype=torch.float32, device=g.device)
    check(lib.fil_embed_segment_sum(ptr(g), ptr(perm), ptr(starts), ptr(rows), ptr(values), None, rows.numel(), K, stream_ptr()),
          "fil_embed_segment_sum")
    # the skipped bucket (id -1: out-of-range ids / frozen fields) sorts first and its value row is never written: drop it, so
    # that `rows` really are the unique touched rows (a lazy / sparse optimizer must not see a spurious row 0 every step).
    # Only possible with a range check or frozen fields; unique_consecutive above has synchronised already.
    if (sizes is not None or frozen is not None) and rows.numel() > 0 and bool(rows[0] < 0):
        rows, values = rows[1:], values[1:]
    return rows, values


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, offsets, sizes, idx, frozen, sparse_grad, atomic, oob_count, layout_key=None, xt_out=None, out_dtype=None):
        _require_cuda(table, offsets, idx)
        table = _f32c(table)
        idx = idx.to(torch.int64).contiguous()
        offsets = offsets.to(torch.int64).contiguous()
        B, F = idx.shape
        K = table.shape[1]
        del _SORT_CACHE[:]
        out_dtype = out_dtype or torch.float32
        if out_dtype not in (torch.float32, torch.bfloat16) or (xt_out is not None and out_dtype != torch.float32):
            raise FilError("embed_gather: out_dtype %s (float32, or bfloat16 without emit_xt)" % out_dtype)
        out = torch.empty((B, F, K), dtype=out_dtype, device=table.device)
        if xt_out is not None:     # both layouts in one pass: the packed block and its [B*K, F] transpose (fil.h)
            check(_lib.load().fil_embed_gather_xt(ptr(table), ptr(offsets), ptr(sizes), ptr(idx), ptr(out), ptr(xt_out), ptr(oob_count),
                                                  B, F, K, stream_ptr()), "fil_embed_gather_xt")
        else:
            check(_lib.load().fil_embed_gather_dt(ptr(table), ptr(offsets), ptr(sizes), ptr(idx), ptr(out), ptr(oob_count), B, F, K,
                                                  FIL_F32 if out_dtype == torch.float32 else FIL_BF16, stream_ptr()), "fil_embed_gather_dt")
        ctx.save_for_backward(offsets, idx, *[t for t in (sizes, frozen) if t is not None])
        ctx.cfg = (tuple(table.shape), sizes is not None, frozen is not None, bool(sparse_grad), bool(atomic), layout_key)
        return out

    @staticmethod
    def backward(ctx, g):
        table_shape, has_sizes, has_frozen, sparse_grad, atomic, layout_key = ctx.cfg
        sv = list(ctx.saved_tensors)
        offsets, idx = sv[:2]
        rest = sv[2:]
        sizes = rest.pop(0) if has_sizes else None
        frozen = rest.pop(0) if has_frozen else None
        B, F = idx.shape
        K = table_shape[1]
        # (a bf16 block's gradient arrives as bf16: the dense deterministic path reads it as it is, the others take fp32)
        g_bf16 = g.dtype == torch.bfloat16 and not atomic and not sparse_grad
        g = g.contiguous() if g_bf16 else _f32c(g)
        if atomic:      # opt-in: fp32 atomics into a zeroed dense table (order of additions not fixed)
            dtable = torch.zeros(table_shape, dtype=torch.float32, device=g.device)
            check(_lib.load().fil_embed_scatter_add(ptr(offsets), ptr(sizes), ptr(idx), ptr(g), ptr(dtable), B, F, K, stream_ptr()),
                  "fil_embed_scatter_add")
            if frozen is not None:
                raise FilError("embed_gather: frozen fields are not supported by the atomic scatter-add")
        elif sparse_grad:   # what Keras hands its optimizers (IndexedSlices): only the touched rows exist
            rows, values = embed_grad_rows(offsets, sizes, idx, g, frozen, layout_key, table_shape[0])
            dtable = torch.sparse_coo_tensor(rows.unsqueeze(0), values, table_shape)
        else:               # dense table, deterministic, no data-dependent shapes (HIP-graph capturable)
            lib = _lib.load()
            sorted_ids, perm = _sorted_row_ids(offsets, sizes, frozen, idx, layout_key, table_shape[0], per_field=True)
            dtable = torch.zeros(table_shape, dtype=torch.float32, device=g.device)
            check(lib.fil_embed_run_sum_dt(ptr(g), ptr(perm), ptr(sorted_ids), ptr(dtable), B * F, K, FIL_BF16 if g_bf16 else FIL_F32,
                                           stream_ptr()), "fil_embed_run_sum_dt")
        return dtable, None, None, None, None, None, None, None, None, None, None


def embed_gather(table, offsets, idx, sizes=None, frozen=None, sparse_grad=False, atomic=False, oob_count=None, layout_key=None,
                 emit_xt=False, out_dtype=None):
    """table [sum V_f, K] (all fields concatenated), offsets [F], idx [B,F] -> packed [B,F,K].
    sizes [F] int64: ids outside [0, V_f) give zero rows (counted in oob_count, an int32 device scalar) and no gradient.
    The gradient is deterministic (sorted segment sums); sparse_grad=True returns it as a sparse COO tensor over the touched
    rows instead of a dense table; atomic=True selects the fp32-atomic scatter-add instead."""
    if not emit_xt:   # (out_dtype=torch.bfloat16: the block leaves the gather rounded to bf16 and its gradient is read as bf16 -- no cast launches)
        return _EmbedFn.apply(table, offsets, sizes, idx, frozen, sparse_grad, atomic, oob_count, layout_key, None, out_dtype)
    # emit_xt: the same launch also writes the block transposed to [B*K, F], the layout the CIN kernels read; it rides on the
    # result as `_fil_xt` (the CIN layer picks it up: no second pass over the block)
    B, F = idx.shape
    xt = torch.empty((B * table.shape[1], F), dtype=torch.float32, device=table.device)
    out = _EmbedFn.apply(table, offsets, sizes, idx, frozen, sparse_grad, atomic, oob_count, layout_key, xt)
    out._fil_xt = xt
    out._fil_xt_version = out._version      # a consumer ignores xt once the block has been modified in place
    return out
