"""GPU parity: HIP path (through the C ABI, via ml_function_amd.functional) vs the fp64 oracle.

Tolerance: norm-relative max|a-b| / max|b| <= 1e-5 against the fp64 oracle (BASELINE.json north_star:
"within 1e-5 relative fp32"; element-wise relative error is meaningless because outputs cross zero,
SURVEY.md section 0.7).  Field-index work is bit-exact.
"""
import numpy as np
import pytest
import torch

from ml_function_amd import synth
from oracle import closed

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda")


def rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64) if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


MEASURED = []      # (name, error, bar) of every check() since the last report(): the benchmark-shape tests print theirs


def check(name, got, want, tol=TOL):
    e = rel(got, want)
    MEASURED.append((name, e, tol))
    assert np.isfinite(e) and e <= tol, "%s: rel err %.3e > %.1e" % (name, e, tol)


def report(title):
    print(title + ": " + ", ".join("%s %.1e" % (n.split(" ", 1)[-1], e) for n, e, _ in MEASURED))
    del MEASURED[:]


def oracle_device():
    """Where the fp64 oracle GRAPH (oracle/graph.py: torch ops, dtype- and device-parametric) is evaluated for the B=4096 cases: on
    the GPU in float64 -- torch's own elementwise / rocBLAS fp64 kernels, nothing of libfil_hip.so -- because 4096 samples of the
    materialised outer product are minutes of host CPU (round 5: the suite took 1038 s of the driver's 1200 s limit, most of it here).
    test_oracle_graph_on_the_gpu_agrees_with_the_cpu keeps the GPU evaluation pinned to the CPU one.  FIL_ORACLE_DEVICE=cpu forces
    the host."""
    import os
    return torch.device(os.environ.get("FIL_ORACLE_DEVICE", "cuda"))


@pytest.mark.parametrize("mode", [0, 1])
def test_cin_wide_dynamic_range(mode):
    """Fields and weight rows scaled over six decades: outputs 1e-5, gradients 2e-5 norm-relative to the fp64 oracle."""
    from ml_function_amd import functional as Fn
    B, F, K, conv = 48, 39, 16, [128, 128, 128]
    c = synth.cin_case(B, F, K, conv, dist="normal")
    rng = np.random.default_rng(11)
    c["x"] = (c["x"] * 10.0 ** rng.uniform(-3, 3, size=(1, F, 1))).astype(np.float32)
    c["Ws"] = [(w * 10.0 ** rng.uniform(-3, 3, size=(w.shape[0], 1))).astype(np.float32) for w in c["Ws"]]
    x = dev(c["x"]).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b) for b in c["bs"]]
    dw, db = dev(c["dense_w"]), dev(c["dense_b"])
    out = Fn.cin(x, Ws, bs, dw, db, output_dim=1, mode=mode)
    check("wide out", out, closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"], 1), tol=1e-5)
    out.backward(dev(c["g"]))
    dx, dWs, _, _, _ = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"], 1)
    check("wide dx", x.grad, dx, tol=2e-5)
    for l in range(3):
        check("wide dW%d" % l, Ws[l].grad, dWs[l], tol=2e-5)


def _cin_run(c, mode, with_grad=True):
    from ml_function_amd import functional as Fn
    x = dev(c["x"]).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b) for b in c["bs"]]
    out = Fn.cin(x, Ws, bs, dev(c["dense_w"]), dev(c["dense_b"]), output_dim=1, mode=mode)
    if with_grad:
        out.backward(dev(c["g"]))
    return out.detach(), x.grad, [w.grad for w in Ws]


# ---- dynamic range, subnormals, non-finite inputs: 0 = headline (exact), 8 = FIL_CIN_NOSYM, the general first-layer kernels,
# ---- 66 = FIL_CIN_BF16X3 | TAIL_ALWAYS: the labelled split-bf16 mode of the merged quadratic tail (csrc/cin_qsplit.h) at the SAME bars
SPLIT = 2 | 64     # (TAIL_ALWAYS: small batches reach the merged quadratic tail, whose GEMMs the bit re-routes; F = 39 is in its kernel menu)


def _assert_split_kernels_ran(c, res_split):
    """The bit selects something only inside the split kernels' menu: the same case on the exact merged tail must differ in its
    last bits (other kernels, other summation order) -- a silent fall-back would be bit-identical."""
    out, dx, dWs = _cin_run(c, 64)
    assert not torch.equal(out, res_split[0]) or not torch.equal(dx, res_split[1])


@pytest.mark.parametrize("mode", [0, 8, SPLIT])
@pytest.mark.parametrize("exp10", [-9, -6, -3, 3, 5])
def test_cin_extreme_magnitudes(mode, exp10):
    """Inputs scaled by 10^e (per-field spread of three more decades on top): the feature maps then sit at ~10^(2e), 10^(3e)
    (degree 2, 3, 4 in x) -- 1e-36..1e+30 across the parametrisation, the widest range whose exact result stays inside fp32."""
    B, F, K, conv = 40, 39, 16, [128, 128, 128]
    c = synth.cin_case(B, F, K, conv, dist="normal")
    rng = np.random.default_rng(5)
    c["x"] = (c["x"] * 10.0 ** exp10 * 10.0 ** rng.uniform(-1.5, 1.5, size=(1, F, 1))).astype(np.float32)
    out, dx, dWs = _cin_run(c, mode)
    if mode == SPLIT and exp10 == -3:
        _assert_split_kernels_ran(c, (out, dx, dWs))
    want = closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"], 1)
    gx, gWs, _, _, _ = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"], 1)
    check("magnitude 1e%d out" % exp10, out, want, tol=1e-5)
    check("magnitude 1e%d dx" % exp10, dx, gx, tol=2e-5)
    for l in range(3):
        check("magnitude 1e%d dW%d" % (exp10, l), dWs[l], gWs[l], tol=2e-5)


@pytest.mark.parametrize("mode", [0, 8, SPLIT])
def test_cin_subnormal_inputs(mode):
    """Every third field holds fp32 subnormals (1e-39..1e-44), the rest ordinary values: the subnormal fields' products are
    far below the others' rounding error, so the bar must still be met; and a batch of ONLY subnormals gives the
    bias-only answer (finite, equal to the oracle's at the bar)."""
    B, F, K, conv = (24, 39, 8, [32, 48, 16]) if mode == SPLIT else (24, 12, 8, [32, 48, 16])
    c = synth.cin_case(B, F, K, conv, dist="normal")
    rng = np.random.default_rng(6)
    tiny = (10.0 ** rng.uniform(-44, -39, size=(B, F, K)) * np.sign(c["x"])).astype(np.float32)
    assert (np.abs(tiny[tiny != 0]) < np.finfo(np.float32).tiny).all()
    mixed = c["x"].copy()
    mixed[:, ::3, :] = tiny[:, ::3, :]
    for name, xin in (("mixed", mixed), ("all-subnormal", tiny)):
        cc = dict(c, x=xin)
        out, dx, dWs = _cin_run(cc, mode)
        assert torch.isfinite(out).all() and torch.isfinite(dx).all()
        check("subnormal %s out" % name, out, closed.cin_fwd(xin, c["Ws"], c["bs"], c["dense_w"], c["dense_b"], 1), tol=1e-5)
        if name == "mixed":
            gx, gWs, _, _, _ = closed.cin_bwd(xin, c["Ws"], c["bs"], c["dense_w"], c["g"], 1)
            # gradient entries of the subnormal fields are tiny but not zero; the norm-relative bar covers the tensor
            check("subnormal dx", dx, gx, tol=2e-5)
            for l in range(3):
                check("subnormal dW%d" % l, dWs[l], gWs[l], tol=2e-5)


@pytest.mark.parametrize("mode", [0, 8, SPLIT])
def test_cin_nonfinite_inputs_stay_in_their_samples(mode):
    """+-inf / NaN in single samples: a bad sample poisons its own output and dx rows (and the weight gradients), nothing else --
    the good samples return the same values as a run without the bad ones."""
    B, F, K, conv = (16, 39, 8, [32, 32, 16]) if mode == SPLIT else (16, 10, 8, [32, 32, 16])
    c = synth.cin_case(B, F, K, conv, dist="normal")
    bad = c["x"].copy()
    bad[3, 2, 1] = np.inf
    bad[7, 0, 5] = -np.inf
    bad[11, 4, 0] = np.nan
    cc = dict(c, x=bad)
    out2, dx2, dW2 = _cin_run(cc, mode)
    bad_rows = [3, 7, 11]
    fin = torch.isfinite(out2).reshape(-1).cpu().numpy()
    assert not fin[bad_rows].any() and fin[[i for i in range(B) if i not in bad_rows]].all()
    good = [i for i in range(B) if i not in bad_rows]
    assert torch.isfinite(dx2[good]).all()
    cg = dict(c, x=c["x"][good], g=c["g"][good])
    outg, dxg, _ = _cin_run(cg, mode)
    assert torch.equal(out2[good], outg) and torch.equal(dx2[good], dxg)


@pytest.mark.parametrize("conv", [[64, 32], [128, 128, 128]])
def test_cin_large_batch_rows_beyond_2_pow_21(conv):
    """B*K = 2.4 M rows (the dW kernel's buffer descriptors used to span the whole tensor: 2^21 rows at most).  [64, 32]: the general
    kernels; 3x128: the headline path (quadratic tail, merged weight gradients) -- its buffers, descriptors and row splits past 2^21 rows.
    Size-independent checks: the first samples equal a small run (batch independence), and the weight gradient equals the sum of
    the two half-batch gradients."""
    from ml_function_amd import functional as Fn
    B, F, K = 150000, 39, 16
    c = synth.cin_case(64, F, K, conv, dist="uniform")
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((B, F, K), device="cuda", generator=g) - 0.5)
    x[:64] = dev(c["x"])
    Ws = [dev(w) for w in c["Ws"]]
    bs = [dev(b) for b in c["bs"]]
    dw, db = dev(c["dense_w"]), dev(c["dense_b"])
    gout = torch.randn((B, 1), device="cuda", generator=g)

    def run(xs, gs, mode=0):
        xs = xs.clone().requires_grad_()
        W2 = [w.clone().requires_grad_() for w in Ws]
        out = Fn.cin(xs, W2, bs, dw, db, mode=mode)
        out.backward(gs)
        return out.detach(), xs.grad, [w.grad for w in W2]

    out, dx, dW = run(x, gout)
    if len(conv) == 2:
        out_s, dx_s, _ = run(x[:64], gout[:64], mode=128)   # (128: one wave per row block, the summation order of the large batch)
        assert torch.equal(out[:64], out_s) and torch.equal(dx[:64], dx_s)
    else:
        # (the same tail at 64 rows: 64 = FIL_CIN_TAIL_ALWAYS, 128 = one wave per row block; the wave shape differs from the large
        # batch's 64-row waves, so the sums are reassociated: equal to fp32 rounding, not bit for bit)
        out_s, dx_s, _ = run(x[:64], gout[:64], mode=64 | 128)
        check("large-batch out[:64]", out[:64], out_s.cpu().numpy())
        check("large-batch dx[:64]", dx[:64], dx_s.cpu().numpy())
    h = B // 2
    _, _, dWa = run(x[:h], gout[:h])
    _, _, dWb = run(x[h:], gout[h:])
    for l in range(len(conv)):
        check("large-batch dW%d" % l, dW[l], (dWa[l] + dWb[l]).cpu().numpy(), tol=2e-5)


# ------------------------------------------------------------------ FM
@pytest.mark.parametrize("B,F,K", [(256, 39, 8), (4096, 39, 16), (7, 3, 4), (33, 5, 6), (1, 2, 1), (130, 26, 32)])
@pytest.mark.parametrize("dist", ["uniform", "normal"])
def test_fm(B, F, K, dist):
    from ml_function_amd import functional as Fn
    c = synth.fm_case(B, F, K, dist=dist)
    emb = dev(c["emb"]).requires_grad_()
    lin = dev(c["lin"]).requires_grad_()
    out = Fn.fm(emb, lin)
    check("fm out", out, closed.fm_fwd(c["emb"], c["lin"]))
    out.backward(dev(c["g"]))
    demb, dlin = closed.fm_bwd(c["emb"], c["g"])
    check("fm demb", emb.grad, demb)
    check("fm dlin", lin.grad, dlin)


def test_fm_no_linear_and_bf16():
    from ml_function_amd import functional as Fn
    c = synth.fm_case(512, 39, 16)
    out = Fn.fm(dev(c["emb"]), None)
    check("fm nolin", out, closed.fm_fwd(c["emb"], None))
    # bf16 storage, fp32 accumulate: compare against the oracle evaluated on the bf16-rounded inputs
    eb = dev(c["emb"]).bfloat16().requires_grad_()
    e_r = eb.detach().float().cpu().numpy()
    ob = Fn.fm(eb, dev(c["lin"]))
    assert ob.dtype == torch.bfloat16
    check("fm bf16 out", ob, closed.fm_fwd(e_r, c["lin"]), tol=4e-3)  # one bf16 rounding of the output
    gb = dev(c["g"]).bfloat16()
    ob.backward(gb)
    demb, _ = closed.fm_bwd(e_r, gb.float().cpu().numpy())
    check("fm bf16 demb", eb.grad, demb, tol=4e-3)


def test_fm_kat_ones():
    from ml_function_amd import functional as Fn
    B, F, K = 5, 7, 4
    out = Fn.fm(torch.ones(B, F, K, device="cuda"), torch.ones(B, F, device="cuda"))
    assert torch.equal(out.cpu(), torch.full((B, K), F * (F - 1) / 2 + F))


@pytest.mark.parametrize("B,F,K", [(64, 6, 4), (17, 39, 16), (3, 2, 3)])
def test_fm_pairs(B, F, K):
    from ml_function_amd import functional as Fn
    c = synth.fm_case(B, F, K, dist="normal")
    emb = dev(c["emb"]).requires_grad_()
    p = Fn.fm_pairs(emb)
    want = closed.fm_pairs_fwd(c["emb"])
    assert np.array_equal(p.detach().cpu().numpy(), (c["emb"][:, np.triu_indices(F, 1)[0]] * c["emb"][:, np.triu_indices(F, 1)[1]]))
    check("pairs", p, want)
    gp = np.random.default_rng(5).standard_normal(want.shape).astype(np.float32)
    p.backward(dev(gp))
    check("pairs demb", emb.grad, closed.fm_pairs_bwd(c["emb"], gp))


# ------------------------------------------------------------------ DCN
# (8, 6400, 8) ... (300, 5000, 16): outside the register-resident kernels (D > 4096, L > 6, or w and b beyond the LDS): the generic two-pass pair
@pytest.mark.parametrize("B,D,L", [(8192, 1248, 3), (64, 1248, 3), (33, 845, 3), (5, 7, 1), (257, 512, 6), (16, 3200, 2), (9, 130, 4),
                                   (8, 6400, 8), (5, 130, 7), (4, 4100, 2), (6, 4096, 6), (6, 3416, 6), (300, 5000, 16), (1, 9000, 1)])
@pytest.mark.parametrize("dist", ["uniform", "normal"])
def test_dcn(B, D, L, dist):
    from ml_function_amd import functional as Fn
    if dist == "normal" and D > 2000:
        pytest.skip("N(0,1) inputs at D>2000 blow up the cross recurrence")
    c = synth.dcn_case(B, D, L, dist=dist)
    if dist == "normal":
        c["x"] = (c["x"] / np.sqrt(D)).astype(np.float32)  # keep x.w O(1) so three layers stay finite
    x, w, b = dev(c["x"]).requires_grad_(), dev(c["w"]).requires_grad_(), dev(c["b"]).requires_grad_()
    y = Fn.dcn_cross(x, w, b)
    yc, _ = closed.dcn_fwd(c["x"], c["w"], c["b"])
    check("dcn y", y, yc)
    y.backward(dev(c["g"]))
    dx, dw, db = closed.dcn_bwd(c["x"], c["w"], c["b"], c["g"])
    check("dcn dx", x.grad, dx)
    check("dcn dw", w.grad, dw)
    check("dcn db", b.grad, db)


def test_dcn_repeatable():
    from ml_function_amd import functional as Fn
    c = synth.dcn_case(1024, 1248, 3)
    outs = []
    for _ in range(2):
        x, w, b = dev(c["x"]).requires_grad_(), dev(c["w"]).requires_grad_(), dev(c["b"]).requires_grad_()
        Fn.dcn_cross(x, w, b).backward(dev(c["g"]))
        outs.append((x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


# ------------------------------------------------------------------ CIN
CIN_SHAPES = [
    (8, 39, 16, [128]),            # one layer: forward GEMM + head
    (8, 39, 16, [128, 128]),
    (64, 39, 16, [128, 128, 128]),  # north-star architecture, small batch
    (5, 3, 4, [5]),
    (9, 5, 8, [6, 7]),
    (3, 4, 2, [3, 3, 3, 3]),       # K not a multiple of 4
    (7, 6, 5, [40, 33]),           # K=5, H not multiples of 32
    (16, 26, 16, [200, 200]),      # reference default width: H > 128 -> two column chunks
    (130, 39, 16, [32, 64]),       # M not a multiple of the 128-row tile
    (4, 64, 4, [16, 16]),          # F at the limit
    (6, 1, 4, [3, 2]),             # one field: the pair reduction of the first layer degenerates to (0,0)
    (6, 2, 4, [3, 2]),             # even F: the d = F/2 pairs are met from both ends (half weights)
    (10, 13, 8, [130, 20]),        # odd F, first layer H > 128
    (33, 38, 16, [64, 48, 8]),     # even F near the north-star size
    (3, 5, 130, [6, 7]),           # L*K = 260 pooled columns: the head kernels loop over them
]


@pytest.mark.parametrize("B,F,K,conv", CIN_SHAPES)
# mode 0: shortcuts (+ the reduction split of small batches), 1: general GEMM kernels, 128: shortcuts, one wave per row block
@pytest.mark.parametrize("output_dim,mode", [(1, 0), (2, 0), (1, 1), (2, 1), (1, 128)])
def test_cin(B, F, K, conv, output_dim, mode):
    from ml_function_amd import functional as Fn
    c = synth.cin_case(B, F, K, conv, dist="uniform", output_dim=output_dim)
    # scale the inputs up so the deeper layers are not vanishingly small next to the shallow ones
    c["x"] = (c["x"] * 10).astype(np.float32)
    x = dev(c["x"]).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b).requires_grad_() for b in c["bs"]]
    dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
    out = Fn.cin(x, Ws, bs, dw, db, output_dim=output_dim, mode=mode)
    want = closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"], output_dim)
    check("cin out", out, want)
    out.backward(dev(c["g"]))
    dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"], output_dim)
    check("cin dx", x.grad, dx)
    for l in range(len(conv)):
        check("cin dW%d" % l, Ws[l].grad, dWs[l])
        check("cin db%d" % l, bs[l].grad, dbs[l])
    if output_dim == 1:
        check("cin ddense_w", dw.grad, ddw)
        check("cin ddense_b", db.grad, ddb)


# fused tail (csrc/cin_tail.h): the last two layers as one implicit GEMM with F+2 columns.  Mode 64 = use it wherever it is
# defined (L >= 3, F <= 62), also at shapes where it saves nothing -- every instantiation family is reached at a small size.
TAIL_SHAPES = [
    (64, 39, 16, [128, 128, 128]),   # the north-star architecture (the default mode picks the tail here by itself)
    (9, 5, 8, [6, 7, 5]),            # one 16-column block, 2 steps per h
    (3, 4, 2, [3, 3, 3, 3]),         # four layers: two general layers below the tail
    (33, 38, 16, [64, 48, 8]),       # even F near the north-star size
    (17, 40, 4, [20, 24, 16]),       # F = 40: F+1 = 41 dZ reduction columns -> one more float4 per tile than JT/4
    (10, 13, 8, [130, 20, 9]),       # the map below the tail has two column chunks (row stride 256)
    (6, 16, 4, [10, 200, 12]),       # H_p = 200 > 128
    (5, 26, 4, [8, 12, 10]),         # 7 steps per h, two column blocks
    (4, 47, 2, [5, 6, 7]),           # four column blocks, 12 steps per h
    (4, 62, 2, [5, 6, 7]),           # F at the tail's limit
    (4, 64, 2, [3, 3, 3]),           # beyond it: falls back to the last-layer shortcut alone
    (130, 39, 16, [32, 64, 16]),     # M not a multiple of the row tiles
    (6, 1, 4, [3, 2, 2]),            # one field
    (6, 2, 4, [3, 2, 4]),
    (300, 21, 8, [16, 16, 16]),      # several workgroups, several row splits of the weight-gradient kernel
    (100, 5, 8, [6, 7, 5]),          # M = 800 rows: four 256-row blocks of dc partials on a net whose parameter kernel needs < 1 KB of LDS
    (73, 6, 6, [8, 8, 8]),           # K = 6: samples straddle the 256-row blocks of the operand-row kernel (the head's block partials) and the
                                     # 32-row waves of the forward (no head in its epilogue: the separate head kernel runs)
    (21, 7, 5, [9, 8, 6]),           # K = 5, odd everything
    (10, 5, 64, [6, 7, 5]),          # K = 64 > 32: a sample is two waves of the forward (B = 9 draws sum_m dP_L = -0.002 against 46 of
                                     # absolute terms: db of the last layer is then a cancellation test, not a kernel test)
]
TAIL_VARIANT_SHAPES = [TAIL_SHAPES[i] for i in (0, 1, 3, 6, 14)]


def _tail_case(B, F, K, conv, output_dim, mode):
    from ml_function_amd import functional as Fn
    c = synth.cin_case(B, F, K, conv, dist="uniform", output_dim=output_dim)
    c["x"] = (c["x"] * 10).astype(np.float32)
    c["bs"] = [(0.1 * np.random.default_rng(l).standard_normal(b.shape)).astype(np.float32) for l, b in enumerate(c["bs"])]   # biases matter here
    x = dev(c["x"]).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b).requires_grad_() for b in c["bs"]]
    dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
    out = Fn.cin(x, Ws, bs, dw, db, output_dim=output_dim, mode=mode)
    check("tail out", out, closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"], output_dim))
    out.backward(dev(c["g"]))
    dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"], output_dim)
    check("tail dx", x.grad, dx)
    for l in range(len(conv)):
        check("tail dW%d" % l, Ws[l].grad, dWs[l])
        check("tail db%d" % l, bs[l].grad, dbs[l])
    if output_dim == 1:
        check("tail ddense_w", dw.grad, ddw)
        check("tail ddense_b", db.grad, ddb)


@pytest.mark.parametrize("B,F,K,conv", TAIL_SHAPES)
# 64: three-layer nets take the quadratic tail with merged weight gradients (cin_qtail.h, cin_qmerge.h), the others the fused tail;
# | 512: FIL_CIN_NOQMERGE -- the quadratic tail with two weight-gradient launches; | 256: FIL_CIN_NOQTAIL -- the F+1-column fused tail
# | 2: FIL_CIN_BF16X3 -- the merged quadratic tail's GEMMs on split-bf16 operands where such kernels exist (else the exact ones)
@pytest.mark.parametrize("output_dim,mode", [(1, 64), (2, 64), (1, 64 | 512), (1, 64 | 256), (1, 64 | 2), (2, 64 | 2)])
def test_cin_fused_tail(B, F, K, conv, output_dim, mode):
    _tail_case(B, F, K, conv, output_dim, mode)


@pytest.mark.parametrize("B,F,K,conv", TAIL_VARIANT_SHAPES)
def test_cin_fused_tail_pooled_output_without_the_quadratic_tail(B, F, K, conv):
    """output_dim = 2 (the pooled blocks are the output, no dense head) on the F+1-column fused tail (FIL_CIN_NOQTAIL): the column of
    test_cin_fused_tail that round 5 pruned, kept on a subset of the shapes."""
    _tail_case(B, F, K, conv, 2, 64 | 256)


@pytest.mark.parametrize("B,F,K,conv", TAIL_VARIANT_SHAPES)
# launch-shape variants on a subset of the shapes: | 4 the 64-row waves large batches get, | 128 one wave per row block instead of the
# small-batch reduction split, | 8 no pair-symmetric kernels (the fused tail with a general first layer)
@pytest.mark.parametrize("mode", [64 | 4, 64 | 128, 64 | 256 | 4, 64 | 512 | 4, 64 | 8])
def test_cin_fused_tail_launch_variants(B, F, K, conv, mode):
    _tail_case(B, F, K, conv, 1, mode)


def _graph_oracle_cin(c, output_dim=1):
    """Every output and gradient of a CIN case from the op-for-op graph (oracle/graph.py:cin, Z materialised) under autograd, in float64
    on oracle_device(), in shards of 512 samples (rows are independent; parameter .grad accumulates over the shards)."""
    from oracle import graph
    B = c["x"].shape[0]
    od = oracle_device()
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=od)
    N = lambda t: t.detach().cpu().numpy()
    Ws, bs, dw, db = [T(w) for w in c["Ws"]], [T(b) for b in c["bs"]], T(c["dense_w"]), T(c["dense_b"])
    for p in Ws + bs + ([dw, db] if output_dim == 1 else []):
        p.requires_grad_()
    outs, dxs = [], []
    shard = 512 if od.type == "cuda" else 256       # (layer 2's outer product of 512 samples: 0.33 GB in fp64)
    for lo in range(0, B, shard):
        x = T(c["x"][lo:lo + shard]).requires_grad_()
        out = graph.cin(x, Ws, bs, dw, db, output_dim=output_dim)
        out.backward(T(c["g"][lo:lo + shard]))
        outs.append(N(out))
        dxs.append(N(x.grad))
    res = dict(out=np.concatenate(outs), dx=np.concatenate(dxs), dW=[N(w.grad) for w in Ws], db=[N(b.grad) for b in bs],
               ddw=N(dw.grad) if output_dim == 1 else None, ddb=N(db.grad) if output_dim == 1 else None)
    if od.type == "cuda":
        torch.cuda.empty_cache()
    return res


# (2048: M = 32,768 rows -- above the quadratic tail's size rule, below the 64-row-wave threshold: its kernels at 32 rows per wave)
@pytest.mark.parametrize("mode,B", [(0, 128), (0, 512), (0, 1024), (0, 2048), (128, 128), (128, 512), (128, 1024)])
def test_cin_small_batches_of_the_benchmark_shape(B, mode):
    """A strong-scaling shard of the benchmark (global B = 4096 over 8 / 32 GPUs: 512 / 128 samples, and 1024): mode 0 splits the
    reduction of every row-parallel kernel over the four waves of a workgroup (fil.h FIL_CIN_NOKSPLIT), mode 128
    keeps one wave per row block -- both against the fp64 oracle, and against each other to fp32 reassociation."""
    from ml_function_amd import functional as Fn
    c = synth.cin_case(B, 39, 16, [128, 128, 128])
    x = dev(c["x"]).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b).requires_grad_() for b in c["bs"]]
    dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
    out = Fn.cin(x, Ws, bs, dw, db, mode=mode)
    if B <= 128:      # the closed-form NumPy oracle (minutes of host CPU at the larger sizes: those take the graph oracle in fp64 on the GPU)
        want = closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"])
        dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"])
    else:
        o = _graph_oracle_cin(c)
        want, dx, dWs, dbs, ddw = o["out"], o["dx"], o["dW"], o["db"], o["ddw"]
    check("shard out", out, want)
    out.backward(dev(c["g"]))
    check("shard dx", x.grad, dx, tol=2e-5)
    for l in range(3):
        check("shard dW%d" % l, Ws[l].grad, dWs[l], tol=2e-5)
        check("shard db%d" % l, bs[l].grad, dbs[l], tol=2e-5)
    check("shard ddense_w", dw.grad, ddw, tol=2e-5)
    if mode == 0 and B <= 1024:     # (the reduction split is used up to 16,384 rows)
        other = Fn.cin(dev(c["x"]), [dev(w) for w in c["Ws"]], [dev(b) for b in c["bs"]], dev(c["dense_w"]), dev(c["dense_b"]), mode=128)
        assert not torch.equal(out, other) and rel(out, other.detach().cpu().numpy()) < 1e-5      # other kernels ran, same function


def test_cin_fused_tail_is_the_default_where_it_pays():
    """Mode 0 picks the fused tail at the north-star architecture (F+2 = 41 -> 48 columns against H = 128) and the result
    differs from the round-2 path (mode 32: last-layer shortcut only) by fp32 reassociation only -- not bit-identical, which
    proves the other kernels ran -- while a narrow net (H = 16 < 48) keeps the round-2 path bit for bit."""
    from ml_function_amd import functional as Fn
    for conv, same in (([128, 128, 128], False), ([16, 16, 16], True)):
        c = synth.cin_case(32, 39, 16, conv)
        args = [dev(c["x"]), [dev(w) for w in c["Ws"]], [dev(b) for b in c["bs"]], dev(c["dense_w"]), dev(c["dense_b"])]
        a, b2, t = Fn.cin(*args, mode=0), Fn.cin(*args, mode=32), Fn.cin(*args, mode=64)
        assert torch.equal(a, b2) == same
        assert rel(a, b2.cpu().numpy()) < 1e-5
        # three layers: mode 64 is the quadratic tail (at any size), 64 | 256 the F+1-column fused tail -- different kernels, same
        # function; at this small batch (M = 512 rows <= 16,384) the default of a wide net is the fused tail, bit for bit
        u = Fn.cin(*args, mode=64 | 256)
        assert not torch.equal(t, u) and rel(t, u.cpu().numpy()) < 1e-5
        assert torch.equal(a, u) == (not same)
        assert not torch.equal(a, t)


_BENCH_ORACLE = {}


def _bench_case(variant):
    """variant "bench": bench.py's own seeded inputs.  "deep": the same net with the inputs x10 and biases of size 0.1, output_dim = 2
    (the pooled blocks themselves are the output): layer 3's pool is then ~1e-2..1e-1 of layer 1's instead of ~5e-3, and every
    block is checked on its own, so a forward error of the deep layers cannot hide behind the first layer's magnitude."""
    B, F, K, conv = 4096, 39, 16, [128, 128, 128]
    if variant == "bench":
        return synth.cin_case(B, F, K, conv), 1
    c = synth.cin_case(B, F, K, conv, output_dim=2)
    c["x"] = (c["x"] * 10).astype(np.float32)
    c["bs"] = [(0.1 * np.random.default_rng(l).standard_normal(b.shape)).astype(np.float32) for l, b in enumerate(c["bs"])]
    return c, 2


def _bench_shape_oracle(variant="bench"):
    """fp64 oracle of the benchmark's workload (B=4096, F=39, K=16, 3x128): the op-for-op graph under autograd in shards
    of 512 samples (rows are independent; parameter gradients add), in float64 on oracle_device(), computed once per test session."""
    if variant not in _BENCH_ORACLE:
        c, output_dim = _bench_case(variant)
        _BENCH_ORACLE[variant] = dict(c=c, **_graph_oracle_cin(c, output_dim))
    return _BENCH_ORACLE[variant]


def test_oracle_graph_on_the_gpu_agrees_with_the_cpu():
    """The B=4096 cases take their fp64 oracle from oracle/graph.py evaluated on the GPU (oracle_device()).  This keeps that
    evaluation honest: the same graph on the host CPU, B=64 of the benchmark's CIN net and B=8 of the c5 attention stack, every
    output and gradient to 1e-12 -- fp64 on either device, far below every parity bar."""
    from oracle import graph
    agree = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
    c = synth.cin_case(64, 39, 16, [128, 128, 128])
    res = {}
    for od in ("cpu", "cuda"):
        T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=od).requires_grad_()
        leaves = [T(c["x"])] + [T(w) for w in c["Ws"]] + [T(b) for b in c["bs"]] + [T(c["dense_w"]), T(c["dense_b"])]
        out = graph.cin(leaves[0], leaves[1:4], leaves[4:7], leaves[7], leaves[8], output_dim=1)
        out.backward(torch.tensor(c["g"], dtype=torch.float64, device=od))
        res[od] = [out.detach()] + [p.grad for p in leaves]
    worst = max(agree(a, b) for a, b in zip(res["cuda"], res["cpu"]))
    a = synth.attn_stack_case(8, 200, 16, 4, 16, 3, dist="normal", beta_shift=4.0, center_upper=True)
    for od in ("cpu", "cuda"):
        T = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64, device=od).requires_grad_()
        x, layers = T(a["x"]), [tuple(T(p) for p in lay) for lay in a["layers"]]
        y = graph.autoint_stack(x, layers)
        y.backward(torch.tensor(a["dy"], dtype=torch.float64, device=od))
        res[od] = [y.detach(), x.grad] + [p.grad for lay in layers for p in lay]
    worst = max(worst, max(agree(g, h) for g, h in zip(res["cuda"], res["cpu"])))
    print("oracle graph, GPU fp64 vs CPU fp64: worst disagreement %.1e" % worst)
    assert worst < 1e-12


def _run_bench_shape(c, output_dim, mode, reps=1):
    """The HIP path on `reps` copies of the case's batch (copy i > 0 with the sample order reversed)."""
    from ml_function_amd import functional as Fn
    xs, gs = c["x"], c["g"]
    if reps > 1:
        xs = np.concatenate([xs if i % 2 == 0 else xs[::-1] for i in range(reps)])
        gs = np.concatenate([gs if i % 2 == 0 else gs[::-1] for i in range(reps)])
    x = dev(xs).requires_grad_()
    Ws = [dev(w).requires_grad_() for w in c["Ws"]]
    bs = [dev(b).requires_grad_() for b in c["bs"]]
    dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
    out = Fn.cin(x, Ws, bs, dw, db, output_dim=output_dim, mode=mode)
    out.backward(dev(gs))
    return out, x.grad, [w.grad for w in Ws], [b.grad for b in bs], dw.grad, db.grad


# 0 = headline (quadratic tail, merged weight gradients), 512 = quadratic tail with two weight-gradient launches, 256 = F+1-column
# fused tail, 32 = last-layer shortcut only (the round-2 headline), 1 = general kernels for every layer, 2 = FIL_CIN_BF16X3: the
# headline's three GEMM launches on split-bf16 operands (the labelled mode), held to the SAME bars
@pytest.mark.parametrize("mode", [0, 1, 32, 256, 512, 2])
def test_cin_at_the_benchmark_shape(mode):
    """The launch configuration bench.py times (M = B*K = 65,536 rows: 64-row waves, the weight-gradient split plans and XCD mapping
    of that size, the tails' kernels) against the fp64 oracle -- every output and every gradient, all 4096 samples."""
    o = _bench_shape_oracle()
    del MEASURED[:]
    out, dx, dW, dbias, ddw, ddb = _run_bench_shape(o["c"], 1, mode)
    check("bench-shape out", out, o["out"])
    check("bench-shape dx", dx, o["dx"])
    for l in range(3):
        check("bench-shape dW%d" % l, dW[l], o["dW"][l])
        check("bench-shape db%d" % l, dbias[l], o["db"][l])
    check("bench-shape ddense_w", ddw, o["ddw"])
    check("bench-shape ddense_b", ddb, o["ddb"])
    report("c4 at B=4096, mode %d (bar 1e-5 on everything)" % mode)


@pytest.mark.parametrize("mode", [0, 512, 256, 1, 2])
def test_cin_at_the_benchmark_shape_deep_layers(mode):
    """B = 4096 with inputs x10, biases ~0.1 and output_dim = 2: each layer's pooled block [B, K] is compared with the fp64 oracle ON ITS
    OWN NORM (a 0.1 % error in the third layer's forward is 1e-3 here, not 5e-6 of the whole output), and so is every gradient."""
    o = _bench_shape_oracle("deep")
    del MEASURED[:]
    out, dx, dW, dbias, _, _ = _run_bench_shape(o["c"], 2, mode)
    K = 16
    gt = 1e-5 if mode in (0, 512, 2) else 2e-5      # north_star's bar on the paths the benchmark can take; the comparison paths (F+1-column
    for l in range(3):                           # fused tail, general kernels) measure 1.4e-5 on one bias gradient of these x10 inputs
        check("deep pooled block %d" % l, out[:, l * K:(l + 1) * K], o["out"][:, l * K:(l + 1) * K])
    check("deep dx", dx, o["dx"], tol=gt)
    for l in range(3):
        check("deep dW%d" % l, dW[l], o["dW"][l], tol=gt)
        check("deep db%d" % l, dbias[l], o["db"][l], tol=gt)
    report("c4 deep layers at B=4096, mode %d (bar %.0e)" % (mode, gt))


@pytest.mark.parametrize("mode", [0, 512, 2])
def test_cin_at_twice_the_benchmark_batch(mode):
    """B = 8192 (131,072 rows: the next launch configuration up -- more row splits, two workgroup rounds) = the benchmark batch followed
    by the same samples in reverse order, against the SAME fp64 oracle: outputs and dx repeat, parameter gradients double."""
    o = _bench_shape_oracle()
    del MEASURED[:]
    out, dx, dW, dbias, ddw, ddb = _run_bench_shape(o["c"], 1, mode, reps=2)
    check("2x out", out, np.concatenate([o["out"], o["out"][::-1]]))
    check("2x dx", dx, np.concatenate([o["dx"], o["dx"][::-1]]))
    for l in range(3):
        check("2x dW%d" % l, dW[l], 2 * o["dW"][l])
        check("2x db%d" % l, dbias[l], 2 * o["db"][l])
    check("2x ddense_w", ddw, 2 * o["ddw"])
    check("2x ddense_b", ddb, 2 * o["ddb"])
    report("c4 at B=8192, mode %d (bar 1e-5)" % mode)


@pytest.mark.parametrize("B,F,K,conv", [(64, 39, 16, [128, 128, 128]), (33, 38, 16, [64, 48, 8]), (9, 5, 8, [6, 7]), (16, 26, 16, [200, 200])])
@pytest.mark.parametrize("mode", [0, 1])
def test_cin_wide_wave_instantiations_at_small_sizes(B, F, K, conv, mode):
    """FIL_CIN_MB2 forces the 64-row-per-wave instantiations (what M >= 49,152 rows selects by itself, i.e. the kernels the
    benchmark runs) at sizes the oracle checks in full; FIL_CIN_NOSYM the general first-layer kernels.  Both against the
    fp64 oracle at the usual bars."""
    from ml_function_amd import functional as Fn
    c = synth.cin_case(B, F, K, conv, dist="uniform")
    c["x"] = (c["x"] * 10).astype(np.float32)
    want = closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"])
    dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"])
    for extra in (4, 8, 12):    # MB2, NOSYM, both
        x = dev(c["x"]).requires_grad_()
        Ws = [dev(w).requires_grad_() for w in c["Ws"]]
        bs = [dev(b).requires_grad_() for b in c["bs"]]
        out = Fn.cin(x, Ws, bs, dev(c["dense_w"]), dev(c["dense_b"]), mode=mode | extra)
        check("out (mode %d)" % (mode | extra), out, want)
        out.backward(dev(c["g"]))
        check("dx (mode %d)" % (mode | extra), x.grad, dx, tol=2e-5)
        for l in range(len(conv)):
            check("dW%d (mode %d)" % (l, mode | extra), Ws[l].grad, dWs[l], tol=2e-5)
            check("db%d (mode %d)" % (l, mode | extra), bs[l].grad, dbs[l], tol=2e-5)


def test_cin_unscaled_inputs_match_fp32_reference_accuracy():
    """x ~ U(-0.05,0.05): the outputs are dominated by the biases, which is where a bias-seeded accumulator loses
    accuracy.  The HIP path must stay within a small factor of the fp32 reference graph's own error."""
    from ml_function_amd import functional as Fn
    from oracle import graph
    c = synth.cin_case(64, 39, 16, [128, 128, 128])
    want = closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"])
    t32 = lambda a: torch.tensor(a, dtype=torch.float32)
    ref32 = graph.cin(t32(c["x"]), [t32(w) for w in c["Ws"]], [t32(b) for b in c["bs"]], t32(c["dense_w"]), t32(c["dense_b"]))
    e_ref = rel(ref32, want)
    for mode in (0, 1):
        out = Fn.cin(dev(c["x"]), [dev(w) for w in c["Ws"]], [dev(b) for b in c["bs"]], dev(c["dense_w"]), dev(c["dense_b"]), mode=mode)
        e = rel(out, want)
        assert e <= max(3 * e_ref, 3e-6), (mode, e, e_ref)


def test_cin_kat_ones():
    from ml_function_amd import functional as Fn
    B, F, K = 2, 3, 4
    H1, H2, H3 = 2, 5, 3
    Ws = [torch.ones(F * F, H1, device="cuda"), torch.ones(H1 * F, H2, device="cuda"), torch.ones(H2 * F, H3, device="cuda")]
    bs = [torch.zeros(h, device="cuda") for h in (H1, H2, H3)]
    P = Fn.cin(torch.ones(B, F, K, device="cuda"), Ws, bs, None, None, output_dim=2)
    exp = torch.cat([torch.full((B, K), float(H1 * F ** 2)), torch.full((B, K), float(H2 * H1 * F ** 3)),
                     torch.full((B, K), float(H3 * H1 * H2 * F ** 4))], -1)
    assert torch.equal(P.cpu(), exp)


def test_cin_repeatable_and_batch_independent():
    from ml_function_amd import functional as Fn
    c = synth.cin_case(96, 39, 16, [64, 64])
    def run(sl):
        x = dev(c["x"][sl]).requires_grad_()
        Ws = [dev(w).requires_grad_() for w in c["Ws"]]
        bs = [dev(b).requires_grad_() for b in c["bs"]]
        dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
        out = Fn.cin(x, Ws, bs, dw, db)
        out.backward(dev(c["g"][sl]))
        return out.detach(), x.grad, [w.grad for w in Ws]
    o1, dx1, dW1 = run(slice(None))
    o2, dx2, dW2 = run(slice(None))
    assert torch.equal(o1, o2) and torch.equal(dx1, dx2) and all(torch.equal(a, b) for a, b in zip(dW1, dW2))
    # data-parallel sharding: shard outputs are the rows of the full-batch output; shard dW sum to the full dW
    oa, dxa, dWa = run(slice(0, 48))
    ob, dxb, dWb = run(slice(48, 96))
    assert torch.equal(torch.cat([oa, ob]), o1) and torch.equal(torch.cat([dxa, dxb]), dx1)
    for a, b, f in zip(dWa, dWb, dW1):
        check("shard dW sum", a + b, f.cpu().numpy(), tol=1e-5)


# ------------------------------------------------------------------ field-index work
def test_embed_gather_bit_exact():
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(11)
    vocab = [5, 1000, 37, 2, 64]
    K, B = 16, 333
    tables = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    table = dev(np.concatenate(tables, 0)).requires_grad_()
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device="cuda")
    out = Fn.embed_gather(table, offsets, torch.tensor(idx, device="cuda"))
    assert np.array_equal(out.detach().cpu().numpy(), closed.embed_gather(tables, idx))
    g = rng.standard_normal((B, len(vocab), K)).astype(np.float32)
    out.backward(dev(g))
    want = np.concatenate(closed.embed_scatter_add(idx, g, vocab), 0)
    check("embed dtable", table.grad, want, tol=1e-6)


def test_embed_fields_sharing_a_table_keep_the_global_sort():
    """Two fields looking up ONE table (offsets[f] == offsets[g]) and a layout without `sizes`: the same row then appears under
    two fields, and the per-field sort (fil_embed_sort_fields) would leave it as two runs, of which fil_embed_run_sum keeps one.
    functional._fields_disjoint sends such layouts down the global sort: dense deterministic gradient == the atomic scatter-add ==
    the NumPy sum, and a disjoint layout still takes the per-field sort."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(5)
    K, B = 8, 257
    V = [7, 50, 50, 3]                      # fields 1 and 2 share the 50-row table
    offs = [0, 7, 7, 57]
    table_np = rng.standard_normal((60, K)).astype(np.float32)
    idx_np = np.stack([rng.integers(0, v, B) for v in V], 1)
    g_np = rng.standard_normal((B, 4, K)).astype(np.float32)
    want = np.zeros((60, K), np.float64)
    for f in range(4):
        np.add.at(want, offs[f] + idx_np[:, f], g_np[:, f].astype(np.float64))
    offsets, sizes = torch.tensor(offs, device="cuda"), torch.tensor(V, device="cuda")
    idx, g = torch.tensor(idx_np, device="cuda"), dev(g_np)
    assert not Fn._fields_disjoint(offsets, sizes, None)
    assert not Fn._fields_disjoint(offsets, None, None)                                   # no sizes: ids are unbounded
    assert Fn._fields_disjoint(torch.tensor([0, 7, 57, 107], device="cuda"), sizes, None)
    for kw in (dict(sizes=sizes), dict(), dict(sizes=sizes, atomic=True)):
        table = dev(table_np).requires_grad_()
        Fn.embed_gather(table, offsets, idx, **kw).backward(g)
        check("shared-table embedding gradient %s" % sorted(kw), table.grad, want, tol=1e-6)
    # determinism of the global-sort path on the shared layout
    a, b = [], []
    for dst in (a, b):
        table = dev(table_np).requires_grad_()
        Fn.embed_gather(table, offsets, idx, sizes=sizes).backward(g)
        dst.append(table.grad.clone())
    assert torch.equal(a[0], b[0])


def test_embed_out_of_range_ids_and_determinism():
    """Ids outside a field's vocabulary give ZERO rows and no gradient (never the neighbouring field's table: the tables
    are concatenated); the table gradient is deterministic (sorted segment sums): repeats are bit-identical even with a
    few rows hit thousands of times, and equal the fp64 oracle; the sparse (rows, values) form sums to the dense one."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(12)
    vocab = [5, 300, 37, 2]
    K, B = 16, 4096
    tables = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    idx = np.stack([np.minimum(rng.zipf(1.2, B) - 1, v - 1) for v in vocab], 1)      # heavy duplicates
    bad = idx.copy()
    bad[3, 0] = 5            # == V_0: would read row 0 of field 1
    bad[7, 3] = -1           # negative
    bad[9, 1] = 10 ** 9
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device="cuda")
    sizes = torch.tensor(vocab, device="cuda")
    g = rng.standard_normal((B, len(vocab), K)).astype(np.float32)

    def run(ids, **kw):
        table = dev(np.concatenate(tables, 0)).requires_grad_()
        cnt = torch.zeros((), dtype=torch.int32, device="cuda")
        out = Fn.embed_gather(table, offsets, torch.tensor(ids, device="cuda"), sizes=sizes, oob_count=cnt, **kw)
        out.backward(dev(g))
        return out.detach(), table.grad, int(cnt)

    out, grad, n_bad = run(bad)
    assert n_bad == 3
    want = closed.embed_gather(tables, np.clip(bad, 0, np.array(vocab) - 1))
    for (b, f) in [(3, 0), (7, 3), (9, 1)]:
        want[b, f] = 0.0
    assert np.array_equal(out.cpu().numpy(), want)
    gz = g.copy()
    for (b, f) in [(3, 0), (7, 3), (9, 1)]:
        gz[b, f] = 0.0
    want_g = np.concatenate(closed.embed_scatter_add(np.clip(bad, 0, np.array(vocab) - 1), gz, vocab), 0)
    check("embed grad with oob ids", grad, want_g, tol=1e-6)
    out2, grad2, _ = run(bad)
    assert torch.equal(grad, grad2)                                   # bit-identical repeats
    _, grad_sp, _ = run(bad, sparse_grad=True)
    assert grad_sp.is_sparse and grad_sp._nnz() <= len(np.unique(idx)) * len(vocab) + 8
    check("sparse == dense", grad_sp.to_dense(), grad.cpu().numpy(), tol=1e-6)      # (two fixed summation orders: a lane tree / four accumulators)
    _, grad_at, _ = run(idx, atomic=True)                              # opt-in atomics: same sums, order not fixed
    check("atomic grad", grad_at, np.concatenate(closed.embed_scatter_add(idx, g, vocab), 0), tol=1e-5)
    # frozen field (sparseFea.is_trainable = False): no gradient for its rows
    frozen = torch.tensor([0, 1, 0, 0], dtype=torch.uint8, device="cuda")
    _, grad_fr, _ = run(idx, frozen=frozen)
    lo, hi = vocab[0], vocab[0] + vocab[1]
    assert float(grad_fr[lo:hi].abs().max()) == 0.0 and float(grad_fr[:lo].abs().max()) > 0.0


@pytest.mark.parametrize("B,K", [(4096, 16), (333, 7), (64, 1)])
def test_embed_gather_in_bf16_is_the_rounded_fp32_block_and_takes_a_bf16_gradient(B, K):
    """embed_gather(out_dtype=bfloat16) (fil_embed_gather_dt / fil_embed_run_sum_dt): the block is the fp32 gather rounded to bf16, bit for
    bit; its gradient, arriving as bf16, gives exactly the table gradient of the fp32 path fed the same values (converted on load, summed
    in fp32, in the same order)."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(B + K)
    vocab = [7, 500, 3, 12000]
    F = len(vocab)
    table = dev(rng.standard_normal((sum(vocab), K)).astype(np.float32)).requires_grad_()
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device="cuda")
    sizes = torch.tensor(vocab, device="cuda")
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in vocab], 1), device="cuda")
    g = dev(rng.standard_normal((B, F, K)).astype(np.float32)).bfloat16()
    out32 = Fn.embed_gather(table, offsets, idx, sizes=sizes)
    out16 = Fn.embed_gather(table, offsets, idx, sizes=sizes, out_dtype=torch.bfloat16)
    assert out16.dtype == torch.bfloat16 and torch.equal(out16, out32.detach().bfloat16())
    out16.backward(g)
    g16 = table.grad.clone()
    table.grad = None
    out32.backward(g.float())
    assert torch.equal(g16, table.grad)


@pytest.mark.parametrize("B", [1, 333, 1000, 2000, 4096, 5000, 8192])
def test_embed_sort_fields_is_a_stable_sort_within_each_field(B):
    """fil_embed_sort_fields (one launch, a bitonic network per field in LDS) against torch's stable sort of the same field: the same
    (row id, position) sequence, skipped entries (-1: out of range / frozen field) first."""
    import ctypes
    from ml_function_amd import _lib
    from ml_function_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    rng = np.random.default_rng(B)
    vocab = [7, 50000, 3, 900, 2]
    F = len(vocab)
    idx = np.stack([np.minimum(rng.zipf(1.3, B) - 1, v - 1) for v in vocab], 1).astype(np.int64)
    idx[rng.integers(0, B, max(1, B // 50)), rng.integers(0, F, max(1, B // 50))] = -3       # out of range
    idx[0, 1] = 50000
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device="cuda")
    sizes = torch.tensor(vocab, device="cuda")
    frozen = torch.tensor([0, 0, 0, 1, 0], dtype=torch.uint8, device="cuda")
    ids = torch.tensor(idx, device="cuda")
    s_ids = torch.empty(B * F, dtype=torch.int64, device="cuda")
    perm = torch.empty(B * F, dtype=torch.int64, device="cuda")
    off = offsets.cpu().numpy()
    for max_vocab in (0, sum(vocab)):        # 64-bit composites (bound unknown) and 32-bit ones
        assert lib.fil_embed_sort_fields(ptr(offsets), ptr(sizes), ptr(frozen), ptr(ids), ptr(s_ids), ptr(perm), B, F, max_vocab, stream_ptr()) == 0
        got_ids, got_perm = s_ids.cpu().numpy().reshape(F, B), perm.cpu().numpy().reshape(F, B)
        for f in range(F):
            ok = (idx[:, f] >= 0) & (idx[:, f] < vocab[f]) & (f != 3)
            row = np.where(ok, off[f] + idx[:, f], -1)
            order = np.argsort(row, kind="stable")
            assert np.array_equal(got_ids[f], row[order]) and np.array_equal(got_perm[f], order * F + f)


def test_sparse_embed_layer_options():
    """pre_weight, is_trainable, emb_reg of the sparseFea descriptor (interactive_layer.py:209-218) and the id check."""
    from ml_function_amd import models
    from ml_function_amd.layers import SparseEmbed
    from ml_function_amd.layers.base import collect_regularization_loss
    rng = np.random.default_rng(3)
    info = models.make_sparse_info([6, 9, 4], embed_dim=8)
    w1 = rng.standard_normal((9, 8)).astype(np.float32)
    info[1] = info[1]._replace(pre_weight=[w1], is_trainable=False, emb_reg=0.0)
    info[0] = info[0]._replace(emb_reg=0.5)
    info[2] = info[2]._replace(emb_reg=0.0)
    emb = SparseEmbed(info, use_flatten=False, check_ids=True)
    idx = torch.tensor(np.stack([rng.integers(0, v, 32) for v in (6, 9, 4)], 1), device="cuda")
    outs = emb(idx)
    assert np.array_equal(outs[1][:, 0].detach().cpu().numpy(), w1[idx[:, 1].cpu().numpy()])     # pre_weight rows, bit-exact
    reg = collect_regularization_loss(emb)
    assert abs(float(reg) - 0.5 * float(emb.embeddings[:6].square().sum())) < 1e-6              # l2(emb_reg) on field 0 only
    (torch.cat(outs, 1).sum() + reg).backward()
    gr = emb.embeddings.grad
    assert float(gr[6:15].abs().max()) == 0.0 and float(gr[:6].abs().max()) > 0.0               # frozen field 1
    bad = idx.clone()
    bad[5, 2] = 4
    with pytest.raises(IndexError, match="C3"):
        emb(bad)
    quiet = SparseEmbed(info, use_flatten=False, check_ids=False)
    assert float(quiet(bad)[2][5].abs().max()) == 0.0                                            # Keras-on-GPU semantics


# ------------------------------------------------------------------ AutoInt interacting layer
ATTN_SHAPES = [(4, 200, 16, 4, 16), (3, 39, 16, 3, 8), (2, 5, 4, 2, 4), (5, 17, 8, 1, 16), (2, 33, 24, 2, 16), (64, 39, 16, 2, 16),
               (2, 230, 16, 2, 16)]      # (F > 208: the 32-key-block instantiations, attention_dim == 16 form)


@pytest.mark.parametrize("B,F,K,H,A", ATTN_SHAPES)
@pytest.mark.parametrize("use_res,use_ln", [(True, True), (False, True), (True, False)])
def test_attn_fused(B, F, K, H, A, use_res, use_ln):
    from ml_function_amd import functional as Fn
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    t = {n: dev(c[n]).requires_grad_() for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    y = Fn.autoint_interact(t["x"], t["Wq"], t["Wk"], t["Wr"] if use_res else None,
                            t["gamma"] if use_ln else None, t["beta"] if use_ln else None)
    want = closed.attn_fwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], c["beta"], use_res=use_res, use_ln=use_ln)
    check("attn y", y, want)
    y.backward(dev(c["dy"]))
    dx, dWq, dWk, dWr, dg, db = closed.attn_bwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], c["beta"], c["dy"],
                                                use_res=use_res, use_ln=use_ln)
    # a ReLU/LN kink may flip on an fp32 rounding for isolated elements: gradients are compared norm-relative
    check("attn dx", t["x"].grad, dx, tol=2e-5)
    check("attn dWq", t["Wq"].grad, dWq, tol=2e-5)
    check("attn dWk", t["Wk"].grad, dWk, tol=2e-5)
    if use_res:
        check("attn dWr", t["Wr"].grad, dWr, tol=2e-5)
    if use_ln:
        check("attn dgamma", t["gamma"].grad, dg, tol=2e-5)
        check("attn dbeta", t["beta"].grad, db, tol=2e-5)


@pytest.mark.parametrize("B,F,K,H,A", [(4, 200, 16, 4, 16), (3, 39, 16, 3, 8), (2, 33, 24, 2, 16), (2, 230, 16, 2, 16)])
@pytest.mark.parametrize("fused", [True, False])
def test_attn_f16_mfma_mode(B, F, K, H, A, fused):
    """BASELINE config 5 ("fp16 MFMA QK^T V"): operands of the matrix products rounded to fp16, fp32 accumulation.
    Tolerances of this labelled mode (norm-relative to the fp64 oracle): 5e-3 on outputs, 2e-2 on gradients
    (fp16 has an 11-bit significand: 4.9e-4 per rounded operand; the fp32 mode's bar stays 1e-5)."""
    from ml_function_amd import functional as Fn
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    t = {n: dev(c[n]).requires_grad_() for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    args = (t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"])
    if fused:
        # (a) the layer as AutoInt uses it.  Outputs are compared at 5e-3.  For the gradients the ReLU kink matters:
        # an output within ~1e-3 of zero can land on the other side in the f16 forward, and every such element moves
        # its whole upstream gradient (norm error ~ sqrt(fraction flipped)), so the random case is held to 0.25 ...
        y = Fn.autoint_interact(*args, precision="f16_mfma")
        want = closed.attn_fwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], c["beta"], use_res=True, use_ln=True)
        check("attn f16 y", y, want, tol=5e-3)
        y32 = Fn.autoint_interact(*args)
        assert float((y.detach() - y32.detach()).abs().max()) > 0.0, "the f16 mode must not silently run the fp32 kernels"
        y.backward(dev(c["dy"]))
        grads = closed.attn_bwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], c["beta"], c["dy"], use_res=True, use_ln=True)
        names = ["x", "Wq", "Wk", "Wr", "gamma", "beta"]
        for n, want_g in zip(names, grads):      # smoke only (finite, right scale): the bar that bites is (b)
            assert torch.isfinite(t[n].grad).all()
            check("attn f16 d" + n, t[n].grad, want_g, tol=0.25)
        # ... (b) and with beta shifted so that no output sits near the kink, the same backward is held to 2e-2
        beta_hi = (c["beta"] + 6.0).astype(np.float32)
        for n in names:
            t[n].grad = None
        tb = dev(beta_hi).requires_grad_()
        Fn.autoint_interact(t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], tb, precision="f16_mfma").backward(dev(c["dy"]))
        grads = closed.attn_bwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], beta_hi, c["dy"], use_res=True, use_ln=True)
        for n, got, want_g in zip(names, [t["x"].grad, t["Wq"].grad, t["Wk"].grad, t["Wr"].grad, t["gamma"].grad, tb.grad], grads):
            check("attn f16 (no kink) d" + n, got, want_g, tol=2e-2)
    else:
        av, res = Fn.mult_head_attention(*args, precision="f16_mfma")
        av32, res32 = Fn.mult_head_attention(*args)
        check("attn f16 av", av, av32.detach().cpu().numpy(), tol=5e-3)
        check("attn f16 res", res, res32.detach().cpu().numpy(), tol=5e-3)
        (av.sum() + (res * res).sum()).backward()
        g16 = [t[n].grad.clone() for n in ["x", "Wq", "Wk", "Wr"]]
        for n in t:
            t[n].grad = None
        (av32.sum() + (res32 * res32).sum()).backward()
        for name, a, b in zip(["dx", "dWq", "dWk", "dWr"], g16, [t[n].grad for n in ["x", "Wq", "Wk", "Wr"]]):
            check("attn f16 unfused " + name, a, b.detach().cpu().numpy(), tol=2e-2)


@pytest.mark.parametrize("B,F,K,H,A,precision", [(2, 200, 64, 4, 16, "f32"), (3, 170, 16, 4, 16, "f32"), (5, 200, 64, 4, 16, "f16_mfma"),
                                                  (2, 190, 64, 4, 16, "f16_mfma"), (3, 170, 64, 4, 16, "f16_mfma"),
                                                  (2, 300, 64, 4, 16, "f16_mfma")])    # (the dx image does not fit: global second visit, f16)
def test_attn_two_waves_per_head(B, F, K, H, A, precision):
    """Shapes whose LDS footprint lets one workgroup per CU only: the backward then runs two waves per head over alternate
    query blocks (odd and even block counts, more workgroups than samples).  The f32 cases hold the strict bar, so they
    pin the block/job bookkeeping the f16 instantiations share."""
    from ml_function_amd import functional as Fn
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    beta = (c["beta"] + (6.0 if precision != "f32" else 0.0)).astype(np.float32)     # f16: keep outputs off the ReLU kink
    t = {n: dev(c[n]).requires_grad_() for n in ["x", "Wq", "Wk", "Wr", "gamma"]}
    tb = dev(beta).requires_grad_()
    y = Fn.autoint_interact(t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], tb, precision=precision)
    ty, tg = (1e-5, 2e-5) if precision == "f32" else (5e-3, 2e-2)
    check("attn wph2 y", y, closed.attn_fwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], beta, use_res=True, use_ln=True), tol=ty)
    y.backward(dev(c["dy"]))
    grads = closed.attn_bwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], beta, c["dy"], use_res=True, use_ln=True)
    for n, got, want in zip(["x", "Wq", "Wk", "Wr", "gamma", "beta"],
                            [t["x"].grad, t["Wq"].grad, t["Wk"].grad, t["Wr"].grad, t["gamma"].grad, tb.grad], grads):
        check("attn wph2 d" + n, got, want, tol=tg)


@pytest.mark.parametrize("B,F,Hp,Ap,H,A", [(3, 39, 3, 8, 2, 8), (2, 200, 4, 16, 4, 16), (2, 21, 2, 5, 3, 4)])
@pytest.mark.parametrize("precision", ["f32", "f16_mfma"])
def test_attn_head_major_input(B, F, Hp, Ap, H, A, precision):
    """x given as a previous layer's [H',B,F,A'] output (fil.h: x_chunk = A') == the same layer on the materialised
    head-concat [B,F,H'*A'] (reference ESULayer convention, behavior_layer.py:973), bit for bit, forward and backward;
    the input gradient comes back head-major."""
    from ml_function_amd import functional as Fn
    K = Hp * Ap
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    xh = np.ascontiguousarray(np.transpose(c["x"].reshape(B, F, Hp, Ap), (2, 0, 1, 3)))      # [H',B,F,A']
    assert np.array_equal(closed.head_concat(xh), c["x"])
    outs = []
    for head_major in (False, True):
        t = {n: dev(c[n]).requires_grad_() for n in ["Wq", "Wk", "Wr", "gamma", "beta"]}
        x = dev(xh if head_major else c["x"]).requires_grad_()
        y = Fn.autoint_interact(x, t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"], precision=precision, head_major=head_major)
        y.backward(dev(c["dy"]))
        dx = x.grad if not head_major else x.grad.permute(1, 2, 0, 3).reshape(B, F, K)
        outs.append([y.detach(), dx] + [t[n].grad for n in ["Wq", "Wk", "Wr", "gamma", "beta"]])
    for a, b2 in zip(*outs):
        assert torch.equal(a, b2)
    if precision == "f32":
        check("head-major y", outs[1][0], closed.attn_fwd(c["x"], c["Wq"], c["Wk"], c["Wr"], c["gamma"], c["beta"]))


@pytest.mark.parametrize("B,F,K,H,A,L", [(3, 200, 16, 4, 16, 3), (5, 39, 16, 2, 8, 2), (2, 50, 8, 8, 4, 2)])
def test_attn_stack(B, F, K, H, A, L):
    """Stack of interacting layers (BASELINE config 5: L=3) against the fp64 oracle, fp32 mode.  dWq of the upper layers
    is ill-conditioned (cancellation; torch-CPU fp32 of the oracle graph shows the same 1e-4 error), bar x10 there."""
    from ml_function_amd import functional as Fn
    c = synth.attn_stack_case(B, F, K, H, A, L, dist="normal")
    x = dev(c["x"]).requires_grad_()
    layers = [tuple(dev(p).requires_grad_() for p in lay) for lay in c["layers"]]
    y = Fn.autoint_stack(x, layers)
    check("stack y", y, closed.attn_stack_fwd(c["x"], c["layers"]))
    y.backward(dev(c["dy"]))
    dx, grads = closed.attn_stack_bwd(c["x"], c["layers"], c["dy"])
    check("stack dx", x.grad, dx, tol=5e-5)
    for l in range(L):
        for p, want, n in zip(layers[l], grads[l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            check("stack %s%d" % (n, l), p.grad, want, tol=5e-4 if (n == "dWq" and l > 0) else 5e-5)


_ATTN_BENCH_ORACLE = {}


def _attn_bench_oracle(kink_free):
    """fp64 oracle of config 5 at the size bench.py --workload autoint times (B=4096, F=200, K=16, H=4, A=16, L=3): the
    op-for-op graph (oracle/graph.py:autoint_stack) under autograd in shards of 128 samples -- rows are independent, parameter
    gradients add up over the shards.  kink_free: synth.attn_stack_case(beta_shift=4, center_upper=True), the inputs the
    f16 gradient bars are held on; otherwise bench.py's own seeded inputs (uniform embeddings)."""
    if kink_free not in _ATTN_BENCH_ORACLE:
        from oracle import graph
        B, F, K, H, A, L = 4096, 200, 16, 4, 16, 3
        c = (synth.attn_stack_case(B, F, K, H, A, L, dist="normal", beta_shift=4.0, center_upper=True) if kink_free
             else synth.attn_stack_case(B, F, K, H, A, L))
        od = oracle_device()
        T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=od)
        N = lambda t: t.detach().cpu().numpy()
        layers = [tuple(T(p).requires_grad_() for p in lay) for lay in c["layers"]]
        ys, dxs = [], []
        shard = 256 if od.type == "cuda" else 128      # (the scores of 256 samples: 0.33 GB per layer in fp64)
        for lo in range(0, B, shard):
            x = T(c["x"][lo:lo + shard]).requires_grad_()
            y = graph.autoint_stack(x, layers)
            y.backward(T(c["dy"][:, lo:lo + shard]))
            ys.append(N(y))
            dxs.append(N(x.grad))
        _ATTN_BENCH_ORACLE[kink_free] = dict(c=c, y=np.concatenate(ys, 1), dx=np.concatenate(dxs, 0),
                                             grads=[[N(p.grad) for p in lay] for lay in layers])
        torch.cuda.empty_cache()
    return _ATTN_BENCH_ORACLE[kink_free]


@pytest.mark.parametrize("precision,kink_free", [("f32", False), ("f16_mfma", True), ("f32", True)])
def test_attn_at_the_benchmark_shape(precision, kink_free):
    """BASELINE config 5 at the timed size (B=4096: the persistent-workgroup path, one workgroup accumulating dW* over several
    samples at 13 key tiles; two waves per head for the K=64 layers) against the fp64 oracle: every output and gradient.
      * kink-free inputs (every pre-activation >= 0.6): fp32 mode 1e-5 / 5e-5, the labelled f16-MFMA mode 5e-3 / 2e-2 on EVERY
        gradient of EVERY layer; dWq of the layers above the first is ill-conditioned (the scores of a layer whose input is the
        normalised output of another one barely depend on q: its terms cancel to ~1e-2 of their size, in fp32 just the same --
        tools/diag_attn_f16.py: 4e-5 against 3e-7 for the other gradients), bar x5 there;
      * the benchmark's own inputs, fp32 mode: outputs at 1e-5.  Among the 157 M pre-activations of this batch a few dozen sit
        within an fp32 rounding of zero and land on the other side of the ReLU: such an element switches its whole upstream
        gradient on or off (2.9e-3 of max|dx| for the sample it belongs to; measured: 7 of the 4096 samples).  So dx is held
        sample by sample -- all but 0.5 % of the samples within 5e-5 -- and the parameter gradients, where a flipped element
        shows as ~1e-3 of the (heavily cancelling) sum over 819,200 rows, at 5e-3 (x50 for the ill-conditioned dWq of the upper
        layers).  The bars that bite for the gradients are the kink-free ones above, at this same size."""
    from ml_function_amd import functional as Fn
    o = _attn_bench_oracle(kink_free)
    c = o["c"]
    x = dev(c["x"]).requires_grad_()
    layers = [tuple(dev(p).requires_grad_() for p in lay) for lay in c["layers"]]
    y = Fn.autoint_stack(x, layers, precision=precision)
    ty, tg = (1e-5, 5e-5) if precision == "f32" else (5e-3, 2e-2)
    check("c5 y", y, o["y"], tol=ty)
    y.backward(dev(c["dy"]))
    errs = {}
    if kink_free:
        errs["dx"] = (rel(x.grad, o["dx"]), tg)
    else:
        per_sample = np.abs(x.grad.detach().cpu().numpy().astype(np.float64) - o["dx"]).max((1, 2)) / np.abs(o["dx"]).max()
        errs["dx: fraction of samples off by > 5e-5"] = (float((per_sample > tg).mean()), 5e-3)
        errs["dx: worst sample"] = (float(per_sample.max()), 2e-2)
    for l in range(3):
        for p, want, n in zip(layers[l], o["grads"][l], ["dWq", "dWk", "dWr", "dgamma", "dbeta"]):
            bar = tg if kink_free else 5e-3
            errs["%s%d" % (n, l)] = (rel(p.grad, want), (5 if kink_free else 50) * bar if (n == "dWq" and l > 0) else bar)
    print("c5 at B=4096 %s kink_free=%s: " % (precision, kink_free) + ", ".join("%s %.1e" % (k, v[0]) for k, v in errs.items()))
    bad = {k: v for k, v in errs.items() if not (np.isfinite(v[0]) and v[0] <= v[1])}
    assert not bad, bad


@pytest.mark.parametrize("shape,precision", [((5, 39, 16, 3, 8), "f32"), ((3, 230, 64, 4, 16), "f16_mfma"), ((4, 200, 16, 4, 16), "f16_mfma")])
def test_attn_backward_without_saved_tensors(shape, precision):
    """fil_attn_bwd with av_saved = rstd_saved = y_saved = NULL re-runs the forward into its workspace: same gradients, bit for bit."""
    from ml_function_amd import functional as Fn
    c = synth.attn_case(*shape, dist="normal")
    res = []
    old = Fn._SAVE_AV
    for save in (True, False):
        Fn._SAVE_AV = save
        try:
            t = {n: dev(c[n]).requires_grad_() for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
            Fn.autoint_interact(t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"], precision=precision).backward(dev(c["dy"]))
            res.append([t[n].grad for n in t])
        finally:
            Fn._SAVE_AV = old
    for other in res[1:]:
        for a, b2 in zip(res[0], other):
            assert torch.equal(a, b2)


def test_attn_repeatable_and_batch_independent():
    """Fixed-order reductions: repeated calls are bit-identical; the rows of a sample do not depend on its batch."""
    from ml_function_amd import functional as Fn
    c = synth.attn_case(700, 39, 16, 4, 16, dist="normal")    # more samples than persistent workgroups take one each
    def run(sl):
        t = {n: dev(c[n]).requires_grad_() for n in ["Wq", "Wk", "Wr", "gamma", "beta"]}
        x = dev(c["x"][sl]).requires_grad_()
        y = Fn.autoint_interact(x, t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"], precision="f16_mfma")
        y.backward(dev(c["dy"][:, sl]))
        return [y.detach(), x.grad] + [t[n].grad for n in t]
    a, b2 = run(slice(None)), run(slice(None))
    for u, v in zip(a, b2):
        assert torch.equal(u, v)
    part = run(slice(100, 164))
    assert torch.equal(part[0], a[0][:, 100:164]) and torch.equal(part[1], a[1][100:164])


@pytest.mark.parametrize("lead,Fq,Fk,A,Av", [((3, 2), 39, 39, 8, 8), ((5,), 7, 20, 16, 4), ((2, 4), 200, 200, 16, 16), ((2,), 17, 33, 40, 24)])
@pytest.mark.parametrize("mask_mod", [0, 1, 2])
def test_product_attention_layer(lead, Fq, Fk, A, Av, mask_mod):
    """ProductAttentionLayer.call([q,k,v], mask) on explicit tensors, mask_mod 1 (scores @ mask) and 2 (scores + mask*(-1e5)),
    against the op-for-op oracle (reference behavior_layer.py:292-311) in float64, outputs and all three gradients."""
    from ml_function_amd.layers import ProductAttentionLayer
    from oracle import graph
    rng = np.random.default_rng(21)
    qn, kn = rng.standard_normal(lead + (Fq, A)), rng.standard_normal(lead + (Fk, A))
    vn = rng.standard_normal(lead + (Fk, Av))
    mask = None
    if mask_mod == 1:
        mask = (rng.random(lead[-1:] + (Fk, Fk)) < 0.7).astype(np.float32)        # broadcast over the leading axis
    elif mask_mod == 2:
        mask = (rng.random(lead[-1:] + (1, Fk)) < 0.3).astype(np.float32)         # key padding mask, broadcast over queries
    layer = ProductAttentionLayer(use_scale=True, mask_mod=max(mask_mod, 1))
    q, k, v = [dev(a).requires_grad_() for a in (qn, kn, vn)]
    out = layer([q, k, v], mask=None if mask is None else dev(mask))
    T64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    q64, k64, v64 = T64(qn), T64(kn), T64(vn)
    want = graph.product_attention(q64, k64, v64, use_scale=True, mask=None if mask is None else torch.tensor(mask, dtype=torch.float64),
                                   mask_mod=max(mask_mod, 1))
    check("pattn out", out, want.detach().numpy())
    g = rng.standard_normal(want.shape)
    out.backward(dev(g))
    want.backward(torch.tensor(g))
    check("pattn dq", q.grad, q64.grad.numpy(), tol=2e-5)
    check("pattn dk", k.grad, k64.grad.numpy(), tol=2e-5)
    check("pattn dv", v.grad, v64.grad.numpy(), tol=2e-5)
    if mask_mod == 2:   # masked keys contribute (numerically) nothing
        dead = mask[..., 0, :] > 0
        assert float(v.grad.detach().cpu().numpy()[np.broadcast_to(dead[(None,) * (len(lead) - 1)], lead + (Fk,))].__abs__().max()) < 1e-20


def test_mult_head_attention_with_mask_matches_reference_graph():
    """MultHeadAttentionLayer.call(x, mask) (behavior_layer.py:356-377) with atten_mask_mod 2 against the oracle graph."""
    from ml_function_amd.layers import MultHeadAttentionLayer
    from oracle import graph
    B, F, K, H, A = 3, 39, 16, 3, 8
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    rng = np.random.default_rng(5)
    mask = (rng.random((B, 1, F)) < 0.25).astype(np.float32)
    att = MultHeadAttentionLayer(attention_dim=A, attention_head_dim=H, atten_mask_mod=2)
    x = dev(c["x"]).requires_grad_()
    att._build_device = x.device
    att.build(tuple(c["x"].shape))
    att.built = True
    with torch.no_grad():
        att.query_w.copy_(dev(c["Wq"])); att.key_w.copy_(dev(c["Wk"])); att.res_w.copy_(dev(c["Wr"]))
        att.ln_gamma.copy_(dev(c["gamma"])); att.ln_beta.copy_(dev(c["beta"]))
    av, res = att(x, mask=dev(mask))
    T64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)
    o = {n: T64(c[n]) for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    av_o, res_o = graph.mult_head_attention(o["x"], o["Wq"], o["Wk"], o["Wr"], o["gamma"], o["beta"], mask=torch.tensor(mask, dtype=torch.float64),
                                            mask_mod=2)
    check("masked mha atten_v", av, av_o.detach().numpy())
    check("masked mha res", res, res_o.detach().numpy())
    g1 = rng.standard_normal(av.shape)
    (av * dev(g1)).sum().backward()
    (av_o * torch.tensor(g1)).sum().backward()
    check("masked mha dx", x.grad, o["x"].grad.numpy(), tol=2e-5)
    check("masked mha dWk", att.key_w.grad, o["Wk"].grad.numpy(), tol=2e-5)


def test_attn_unfused_matches_reference_layer_outputs():
    """MultHeadAttentionLayer.call returns [atten_v, res] (behavior_layer.py:377); gradients flow through both."""
    from ml_function_amd import functional as Fn
    from oracle import graph
    B, F, K, H, A = 3, 39, 16, 3, 8
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    t = {n: dev(c[n]).requires_grad_() for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    av, res = Fn.mult_head_attention(t["x"], t["Wq"], t["Wk"], t["Wr"], t["gamma"], t["beta"])
    T64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)
    o = {n: T64(c[n]) for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    av_o, res_o = graph.mult_head_attention(o["x"], o["Wq"], o["Wk"], o["Wr"], o["gamma"], o["beta"])
    check("mha atten_v", av, av_o.detach().numpy())
    check("mha res", res, res_o.detach().numpy())
    rng = np.random.default_rng(9)
    g1, g2 = rng.standard_normal(av.shape), rng.standard_normal(res.shape)
    (av * dev(g1)).sum().add((res * dev(g2)).sum()).backward()
    ((av_o * torch.tensor(g1)).sum() + (res_o * torch.tensor(g2)).sum()).backward()
    for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]:
        check("mha d" + n, t[n].grad, o[n].grad.numpy(), tol=2e-5)


@pytest.mark.parametrize("B,F,K,conv,output_dim", [(6, 80, 4, [5, 4], 1), (5, 9, 4, [300, 6], 2), (3, 4, 2, [2] * 9, 1)])
def test_cin_layer_outside_the_kernel_menu_takes_the_composed_path(B, F, K, conv, output_dim):
    """F > 64, H > 256, L > 8: the reference has no such limits (interactive_layer.py:296-327).  The HIP entry point refuses the shape
    (FIL_ERR_UNSUPPORTED), the CIN layer does not: it runs the reference's op graph with library GEMMs on the GPU."""
    from ml_function_amd import functional as Fn
    from ml_function_amd._lib import FilError
    from ml_function_amd.layers import CIN
    c = synth.cin_case(B, F, K, conv, dist="uniform", output_dim=output_dim)
    c["x"] = (c["x"] * 10).astype(np.float32)
    with pytest.raises(FilError):
        Fn.cin(dev(c["x"]), [dev(w) for w in c["Ws"]], [dev(b) for b in c["bs"]], dev(c["dense_w"]), dev(c["dense_b"]), output_dim=output_dim)
    lay = CIN(conv_size=conv, output_dim=output_dim)
    x = dev(c["x"]).requires_grad_()
    lay(x)            # builds the weights
    with torch.no_grad():
        for l in range(len(conv)):
            lay.conv_kernels[l].copy_(dev(c["Ws"][l])[None])
            lay.conv_biases[l].copy_(dev(c["bs"][l]))
        if output_dim == 1:
            lay.logit_kernel.copy_(dev(c["dense_w"]))
            lay.logit_bias.copy_(dev(c["dense_b"]))
    out = lay(x)
    check("composed cin out", out, closed.cin_fwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["dense_b"], output_dim))
    out.backward(dev(c["g"]))
    dx, dWs, dbs, ddw, ddb = closed.cin_bwd(c["x"], c["Ws"], c["bs"], c["dense_w"], c["g"], output_dim)
    check("composed cin dx", x.grad, dx, tol=2e-5)
    for l in range(len(conv)):
        check("composed cin dW%d" % l, lay.conv_kernels[l].grad[0], dWs[l], tol=2e-5)
    with pytest.raises(FilError):
        lay(torch.tensor(c["x"]))     # a CPU tensor: no CPU path


@pytest.mark.parametrize("B,D,L", [(5, 130, 17), (3, 4100, 20)])
def test_cross_layer_outside_the_kernel_menu_takes_the_composed_path(B, D, L):
    """cross_hidden > 16: the reference has no such limit (interactive_layer.py:255-282).  The HIP entry point refuses the shape
    (FIL_ERR_UNSUPPORTED), the CrossLayer does not: it runs the reference's recurrence with torch ops on the GPU.  (Round 6: D > 4096,
    6 < L <= 16 and parameter sets beyond the LDS are kernel shapes now -- test_dcn.)"""
    from ml_function_amd import functional as Fn
    from ml_function_amd._lib import FilError
    from ml_function_amd.layers import CrossLayer
    c = synth.dcn_case(B, D, L)
    with pytest.raises(FilError):
        Fn.dcn_cross(dev(c["x"]), dev(c["w"]), dev(c["b"]))
    lay = CrossLayer(cross_hidden=L)
    x = dev(c["x"]).requires_grad_()
    lay(x)            # builds the weights
    with torch.no_grad():
        for l in range(L):
            lay.kernel[l].copy_(dev(c["w"][l])[:, None])
            lay.bias[l].copy_(dev(c["b"][l])[:, None])
    y = lay(x)
    assert tuple(y.shape) == (B, D, 1)        # not squeezed, like the reference
    yc, _ = closed.dcn_fwd(c["x"], c["w"], c["b"])
    check("composed dcn y", y[..., 0], yc)
    y.backward(dev(c["g"])[..., None])
    dx, dw, db = closed.dcn_bwd(c["x"], c["w"], c["b"], c["g"])
    check("composed dcn dx", x.grad, dx, tol=2e-5)
    for l in range(L):
        check("composed dcn dw%d" % l, lay.kernel[l].grad[:, 0], dw[l], tol=2e-5)
        check("composed dcn db%d" % l, lay.bias[l].grad[:, 0], db[l], tol=2e-5)
    with pytest.raises(FilError):
        lay(torch.tensor(c["x"]))     # a CPU tensor: no CPU path


def test_autoint_layer_outside_the_kernel_menu_takes_the_composed_path():
    """attention_dim = 32 (the fused kernel's menu ends at A = 16; behavior_layer.py:323-353 has no limit): MultHeadAttentionLayer.call
    and the DnnLayer-fused relu(res + LN(attention)) go through the stand-alone attention kernel and match the oracle graph."""
    from ml_function_amd.layers import MultHeadAttentionLayer
    from oracle import graph
    B, F, K, H, A = 3, 20, 16, 2, 32
    c = synth.attn_case(B, F, K, H, A, dist="normal")
    lay = MultHeadAttentionLayer(A, H)
    x = dev(c["x"]).requires_grad_()
    assert not lay.fits_fused_kernel(x)
    lay(x)
    with torch.no_grad():
        lay.query_w.copy_(dev(c["Wq"])); lay.key_w.copy_(dev(c["Wk"])); lay.res_w.copy_(dev(c["Wr"]))
        lay.ln_gamma.copy_(dev(c["gamma"])); lay.ln_beta.copy_(dev(c["beta"]))
    T64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True)
    o = {n: T64(c[n]) for n in ["x", "Wq", "Wk", "Wr", "gamma", "beta"]}
    av, res = lay(x)
    av_o, res_o = graph.mult_head_attention(o["x"], o["Wq"], o["Wk"], o["Wr"], o["gamma"], o["beta"])
    check("composed mha atten_v", av, av_o.detach().numpy())
    check("composed mha res", res, res_o.detach().numpy())
    y = lay.fused_relu(x)
    y_o = graph.autoint_interacting(o["x"], o["Wq"], o["Wk"], o["Wr"], o["gamma"], o["beta"])
    check("composed autoint y", y, y_o.detach().numpy())
    y.backward(dev(c["dy"]))
    y_o.backward(torch.tensor(c["dy"], dtype=torch.float64))
    check("composed autoint dx", x.grad, o["x"].grad.numpy(), tol=2e-5)
    check("composed autoint dWq", lay.query_w.grad, o["Wq"].grad.numpy(), tol=2e-5)
    check("composed autoint dWk", lay.key_w.grad, o["Wk"].grad.numpy(), tol=2e-5)


def test_cin_step_is_hipgraph_capturable():
    """The C ABI enqueues on the caller's stream without allocating or synchronising, so a whole forward+backward can be
    captured into a HIP graph (torch.cuda.CUDAGraph) and replayed: results equal the eager run bit for bit."""
    from ml_function_amd import functional as Fn
    c = synth.cin_case(64, 39, 16, [128, 128, 128], dist="uniform")
    x, g = dev(c["x"]), dev(c["g"])
    Ws, bs = [dev(w) for w in c["Ws"]], [dev(b) for b in c["bs"]]
    dw, db = dev(c["dense_w"]), dev(c["dense_b"])

    def step():
        out, pooled, saved = Fn.cin_forward_raw(x, Ws, bs, dw, db, 1, 0)
        gr = Fn.cin_backward_raw(x, Ws, bs, dw, pooled, saved, g, 1, 0)
        return out, gr

    out_e, gr_e = step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()  # warm-up on the side stream (workspace growth, lazy module loads) before capture
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g, gr_g = step()
    out_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, out_e)
    assert torch.equal(gr_g["dx"], gr_e["dx"]) and all(torch.equal(a, b) for a, b in zip(gr_g["dW"], gr_e["dW"]))


def test_roctx_ranges_opt_in():
    """FIL_ROCTX=1 (read when the library loads) wraps every profiled scope in a roctx range; the marker library is looked
    up with dlopen.  Without a profiler attached the ranges are no-ops: the layer must run and give the same numbers."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, numpy as np\n"
            "from ml_function_amd import functional as Fn, synth\n"
            "c = synth.cin_case(8, 6, 4, [8, 8], dist='uniform')\n"
            "d = lambda a: torch.tensor(a, device='cuda')\n"
            "out = Fn.cin(d(c['x']), [d(w) for w in c['Ws']], [d(b) for b in c['bs']], d(c['dense_w']), d(c['dense_b']), output_dim=1)\n"
            "print('%.9e' % float(out.double().sum()))\n")
    outs = []
    for flag in ("0", "1"):
        env = dict(os.environ, FIL_ROCTX=flag)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def test_embed_gather_xt_and_cin_transposed_input():
    """N1 fused into its consumer: fil_embed_gather_xt writes the packed block AND its [B*K, F] transpose in one pass (bit-exact
    row copies, out-of-range ids -> zero rows in both), and the CIN kernels read that transpose in place
    (FIL_CIN_X_TRANSPOSED): outputs and every gradient are bit-identical to the path that transposes x itself."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(21)
    vocab = [7, 40, 3, 11, 5, 9]
    B, K, conv = 37, 8, [16, 24, 8]
    F = len(vocab)
    table = dev(rng.standard_normal((sum(vocab), K)).astype(np.float32)).requires_grad_()
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device="cuda")
    sizes = torch.tensor(vocab, device="cuda")
    idx = np.stack([rng.integers(0, v, B) for v in vocab], 1)
    idx[2, 1] = 40          # out of range: zero row in both layouts
    idx_t = torch.tensor(idx, device="cuda")
    cnt = torch.zeros((), dtype=torch.int32, device="cuda")
    blk = Fn.embed_gather(table, offsets, idx_t, sizes=sizes, emit_xt=True, oob_count=cnt)
    ref = Fn.embed_gather(table, offsets, idx_t, sizes=sizes)
    assert int(cnt) == 1 and torch.equal(blk.detach(), ref.detach())
    xt = blk._fil_xt
    assert tuple(xt.shape) == (B * K, F) and torch.equal(xt, ref.detach().permute(0, 2, 1).reshape(B * K, F))
    # an odd embedding width and a single field (scalar path of the plain gather, one LDS row per sample)
    t2 = dev(rng.standard_normal((9, 5)).astype(np.float32))
    off2, i2 = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.tensor(rng.integers(0, 9, (11, 1)), device="cuda")
    b2 = Fn.embed_gather(t2, off2, i2, emit_xt=True)
    assert torch.equal(b2, Fn.embed_gather(t2, off2, i2)) and torch.equal(b2._fil_xt, b2.permute(0, 2, 1).reshape(55, 1))
    c = synth.cin_case(B, F, K, conv, dist="uniform")
    for mode in (0, 1, 64, 64 | 256):      # (64: the quadratic tail, 64 | 256: the fused tail -- both read the transpose in place too)
        res = []
        for use_xt in (False, True):
            x = ref.detach().clone().requires_grad_()
            Ws = [dev(w).requires_grad_() for w in c["Ws"]]
            bs = [dev(b).requires_grad_() for b in c["bs"]]
            dw, db = dev(c["dense_w"]).requires_grad_(), dev(c["dense_b"]).requires_grad_()
            out = Fn.cin(x, Ws, bs, dw, db, output_dim=1, mode=mode, xt=xt if use_xt else None)
            out.backward(dev(c["g"]))
            res.append([out.detach(), x.grad] + [w.grad for w in Ws] + [b.grad for b in bs] + [dw.grad, db.grad])
        for a, b in zip(*res):
            assert torch.equal(a, b)
    # through the layers: SparseEmbed(emit_xt=True) -> CIN picks the transpose up from the re-packed views
    from ml_function_amd.layers import CIN, SparseEmbed
    from ml_function_amd.layers.interactive_layer import pack_fields
    from ml_function_amd.models import make_sparse_info
    info = make_sparse_info(vocab, embed_dim=K)
    outs = []
    for emit in (False, True):
        torch.manual_seed(5)
        emb = SparseEmbed(info, use_flatten=False, emit_xt=emit)
        cin = CIN(conv_size=conv, output_dim=1)
        views = emb(idx_t.clamp(max=2))       # (every id valid for every field)
        if emit:
            assert getattr(views[0]._base, "_fil_xt", None) is not None
        y = cin(pack_fields(views))
        y.sum().backward()
        outs.append((y.detach(), emb.embeddings.grad.clone(), [p.grad.clone() for p in cin.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


# --------------------------------------------------------------------------------------------- score head and loss (N2)
@pytest.mark.parametrize("n_parts,shape", [(1, (7, 1)), (2, (333, 1)), (3, (4096, 1)), (4, (1000,)), (3, (65537, 1))])
def test_score_add_sigmoid_matches_the_oracle(n_parts, shape):
    """ScoreLayer(use_add=True): Add over the parts + sigmoid as one launch each way, against the fp64 graph under autograd."""
    from oracle import graph
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(7 + n_parts)
    parts = [(3.0 * rng.standard_normal(shape)).astype(np.float32) for _ in range(n_parts)]
    g = rng.standard_normal(shape).astype(np.float32)
    tp = [torch.tensor(p, dtype=torch.float64, requires_grad=True) for p in parts]
    want = graph.score_layer(tp, use_add=True)
    want.backward(torch.tensor(g, dtype=torch.float64))
    dp = [dev(p).requires_grad_() for p in parts]
    got = Fn.score_add_sigmoid(dp)
    got.backward(dev(g))
    assert got.shape == tuple(shape)
    check("score", got, want.detach().numpy(), 2e-6)
    for i in range(n_parts):
        check("dpart%d" % i, dp[i].grad, tp[i].grad.numpy(), 2e-6)


def test_score_layer_takes_the_fused_head_and_agrees_with_the_torch_path():
    from ml_function_amd import layers
    rng = np.random.default_rng(11)
    parts = [dev(rng.standard_normal((512, 1))) for _ in range(3)]
    fused = layers.ScoreLayer(use_add=True)(parts)
    plain = torch.sigmoid((parts[0] + parts[1]) + parts[2])
    assert torch.allclose(fused, plain, rtol=0, atol=2e-7)
    # shapes that broadcast (a [B,1] part against a [1] bias) keep the general path
    assert layers.ScoreLayer(use_add=True)([parts[0], dev(np.ones(1))]).shape == (512, 1)
    with pytest.raises(Exception):
        from ml_function_amd import functional as Fn
        Fn.score_add_sigmoid([torch.zeros(4, 1)])          # a CPU tensor: no fallback


@pytest.mark.parametrize("n,eps", [(1, 1e-7), (5, 1e-7), (4096, 1e-7), (4096, 1e-6), (100003, 1e-7)])
def test_binary_crossentropy_matches_the_oracle(n, eps):
    """tf.losses.binary_crossentropy on probabilities (clip, log(p + eps), mean) and its gradient, incl. saturated entries where the
    clip is active (zero gradient); repeats are bit-identical."""
    from oracle import graph
    from ml_function_amd import losses
    rng = np.random.default_rng(n)
    p = rng.uniform(0.001, 0.999, n).astype(np.float32)
    if n >= 5:
        p[:4] = [0.0, 1.0, 1e-9, 0.99999999]
    y = rng.integers(0, 2, n).astype(np.float32)
    tp = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    want = graph.binary_crossentropy(torch.tensor(y, dtype=torch.float64), tp, eps)
    want.backward()
    dpv = dev(p).requires_grad_()
    got = losses.binary_crossentropy(dpv, dev(y), eps=eps)
    (2.0 * got).backward()
    assert got.shape == ()
    check("loss", got.reshape(1), want.detach().numpy().reshape(1), 1e-5)
    inside = (p >= np.float32(eps)) & (p <= np.float32(1) - np.float32(eps))
    gw = 2.0 * tp.grad.numpy()
    ga = dpv.grad.cpu().numpy().astype(np.float64)
    assert np.all(ga[~inside] == 0.0)
    scale = np.abs(gw[inside]).max() if inside.any() else 1.0
    assert np.abs(ga[inside] - gw[inside]).max() <= 1e-5 * scale
    again = losses.binary_crossentropy(dev(p), dev(y), eps=eps)
    assert torch.equal(again, got.detach())


@pytest.mark.parametrize("B,widths,O,dtype", [(4096, (16, 128), 2, "f32"), (4096, (16, 128), 2, "bf16"), (333, (7, 1, 65, 300), 3, "f32"),
                                              (1, (5,), 2, "f32"), (65, (1248, 64), 2, "f32"), (200, (40,), 8, "bf16"),
                                              (4096, (16, 128), 2, "mixed")])
def test_merge_softmax_head_matches_the_oracle(B, widths, O, dtype):
    """MergeScoreLayer's concat -> Dense(softmax) (core_layer.py:86-100; the head of DeepFM / DCN, models.py:87,104) as one launch each
    way (fil_merge_softmax_*) against oracle/graph.py:merge_score_layer in float64: probabilities 1e-6 absolute, every gradient 1e-5
    norm-relative (bf16 storage: the oracle runs on the bf16-rounded parts; dparts come back rounded to bf16: 4e-3), repeat runs
    bit-identical (the weight gradient's block partials are summed in block order by whichever workgroup finishes last)."""
    from ml_function_amd import functional as Fn
    from oracle import graph
    rng = np.random.default_rng(31)
    tdts = [torch.float32 if dtype == "f32" or (dtype == "mixed" and i > 0) else torch.bfloat16 for i in range(len(widths))]   # mixed: DeepFM
    parts_np = [rng.standard_normal((B, w)).astype(np.float32) for w in widths]                                             # under autocast
    parts = [dev(p).to(t).requires_grad_() for p, t in zip(parts_np, tdts)]
    D = sum(widths)
    kernel = dev(rng.standard_normal((D, O)).astype(np.float32) / np.sqrt(D)).requires_grad_()
    bias = dev(rng.standard_normal(O).astype(np.float32) * 0.1).requires_grad_()
    g = rng.standard_normal((B, O)).astype(np.float32)
    out = Fn.merge_softmax(parts, kernel, bias)
    out.backward(dev(g))
    T = lambda t: t.detach().double().cpu().requires_grad_()
    parts64, k64, b64 = [T(p) for p in parts], T(kernel), T(bias)
    want = graph.merge_score_layer(parts64, k64, b64)
    want.backward(torch.tensor(g, dtype=torch.float64))
    assert out.dtype == torch.float32 and tuple(out.shape) == (B, O)
    assert float((out.detach().double().cpu() - want.detach()).abs().max()) < 1e-6
    check("merge head dW", kernel.grad, k64.grad.numpy(), tol=1e-5)
    check("merge head db", bias.grad, b64.grad.numpy(), tol=1e-5)
    for i, (p, p64) in enumerate(zip(parts, parts64)):
        assert p.grad.dtype == tdts[i]
        check("merge head dpart %d" % i, p.grad, p64.grad.numpy(), tol=1e-5 if tdts[i] == torch.float32 else 4e-3)
    first = (out.detach().clone(), kernel.grad.clone(), bias.grad.clone())
    for t in parts + [kernel, bias]:
        t.grad = None
    out2 = Fn.merge_softmax(parts, kernel, bias)
    out2.backward(dev(g))
    assert torch.equal(out2, first[0]) and torch.equal(kernel.grad, first[1]) and torch.equal(bias.grad, first[2])


def test_merge_score_layer_takes_the_fused_head_and_agrees_with_the_torch_path():
    from ml_function_amd.layers import MergeScoreLayer
    rng = np.random.default_rng(3)
    a, b = dev(rng.standard_normal((50, 1, 16)).astype(np.float32)), dev(rng.standard_normal((50, 24)).astype(np.float32))
    lay = MergeScoreLayer()
    first = lay([a, b])                       # builds the Dense: the composed path
    assert lay._fused([a, b]) is not None     # from now on: one launch
    again = lay([a, b])
    assert tuple(again.shape) == (50, 2) and float((again - first).abs().max()) < 1e-6
    assert lay._fused([a.cpu(), b.cpu()]) is None and lay._fused([a.half(), b.half()]) is None
    mixed = lay([a.bfloat16(), b])            # a bf16 part next to an fp32 one (DeepFM under autocast)
    assert mixed.dtype == torch.float32 and float((mixed - first).abs().max()) < 2e-2


@pytest.mark.parametrize("B,I,N,dtype", [(4096, 637, 256, "f32"), (4096, 256, 128, "bf16"), (1, 5, 3, "f32"), (333, 70, 9, "f32"), (130, 64, 200, "bf16")])
def test_dense_relu_layer_matches_float64(B, I, N, dtype):
    """A hidden layer of the zoo's MLPs (DnnLayer, core_layer.py:102-118,201-226: Dense -> skipped residual -> ReLU) as
    functional.dense_relu: library GEMM with the bias + ReLU epilogue forward; ReLU mask + bias gradient as one HIP pass
    (fil_relu_bias_bwd) and two library GEMMs backward.  Against relu(x W + b) in float64 (on the operands as stored): fp32 1e-5, the
    bf16 storage mode 2e-2 (outputs and gradients are ROUNDED to bf16 there); the bias gradient's reduction repeats bit for bit."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(17)
    x = dev(rng.standard_normal((B, I)).astype(np.float32)).requires_grad_()
    W = dev(rng.standard_normal((I, N)).astype(np.float32) / np.sqrt(I)).requires_grad_()
    b = dev(rng.standard_normal(N).astype(np.float32) * 0.3).requires_grad_()
    g = dev(rng.standard_normal((B, N)).astype(np.float32))
    def run():
        for t in (x, W, b):
            t.grad = None
        if dtype == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = Fn.dense_relu(x, W, b)
        else:
            y = Fn.dense_relu(x, W, b)
        y.backward(g.to(y.dtype))
        return y.detach(), x.grad.clone(), W.grad.clone(), b.grad.clone()
    y, dx, dW, db = run()
    assert y.dtype == (torch.float32 if dtype == "f32" else torch.bfloat16) and dx.dtype == W.grad.dtype == b.grad.dtype == torch.float32
    cast = (lambda t: t.detach().double().cpu()) if dtype == "f32" else (lambda t: t.detach().bfloat16().double().cpu())
    x64, W64, b64 = cast(x).requires_grad_(), cast(W).requires_grad_(), cast(b).requires_grad_()
    want = torch.relu(x64 @ W64 + b64)
    want.backward(g.double().cpu() if dtype == "f32" else g.bfloat16().double().cpu())
    tol = 1e-5 if dtype == "f32" else 2e-2
    check("dense_relu y", y, want.detach().numpy(), tol=tol)
    check("dense_relu dx", dx, x64.grad.numpy(), tol=tol)
    check("dense_relu dW", dW, W64.grad.numpy(), tol=tol)
    check("dense_relu db", db, b64.grad.numpy(), tol=tol)
    y2, dx2, dW2, db2 = run()
    assert torch.equal(db, db2) and torch.equal(y, y2)


@pytest.mark.parametrize("B,I,N", [(4096, 64, 1), (4096, 144, 2), (1, 3, 1), (333, 70, 9)])
def test_dense_layer_matches_float64(B, I, N):
    """The logit heads of the zoo's MLPs (Dense(1) / Dense(2), core_layer.py:201-226) as functional.dense: the library's own fp32 GEMM
    forward and for both gradients, column sums for the bias.  Against x W + b in float64: 1e-5; and the Dense layer takes that path
    on the GPU."""
    from ml_function_amd import functional as Fn
    from ml_function_amd.layers.core_layer import Dense
    rng = np.random.default_rng(23 + N)
    x = dev(rng.standard_normal((B, I)).astype(np.float32)).requires_grad_()
    W = dev(rng.standard_normal((I, N)).astype(np.float32) / np.sqrt(I)).requires_grad_()
    b = dev(rng.standard_normal(N).astype(np.float32) * 0.3).requires_grad_()
    g = dev(rng.standard_normal((B, N)).astype(np.float32))
    y = Fn.dense(x, W, b)
    y.backward(g)
    x64, W64, b64 = [t.detach().double().cpu().requires_grad_() for t in (x, W, b)]
    want = x64 @ W64 + b64
    want.backward(g.double().cpu())
    check("dense y", y.detach(), want.detach().numpy(), tol=1e-5)
    check("dense dx", x.grad, x64.grad.numpy(), tol=1e-5)
    check("dense dW", W.grad, W64.grad.numpy(), tol=1e-5)
    check("dense db", b.grad, b64.grad.numpy(), tol=1e-5)
    lay = Dense(N)
    out = lay(x.detach())
    assert type(out.grad_fn).__name__ == "_DenseFnBackward"
    check("Dense layer", out.detach(), (x.detach() @ lay.kernel + lay.bias).detach().double().cpu().numpy(), tol=1e-5)


def test_dnn_layer_takes_the_fused_hidden_layers_and_agrees_with_the_composed_path():
    from ml_function_amd.layers import DnnLayer
    rng = np.random.default_rng(4)
    x = dev(rng.standard_normal((77, 40)).astype(np.float32))
    lay = DnnLayer(hidden_units=[32, 32, 8])       # 40 -> 32 (fused), 32 -> 32 (square: the residual IS added: composed), 32 -> 8 (fused)
    first = lay(x)                                 # builds the Dense layers: the composed path
    assert lay._dense_relu(0, lay.hidden_list[0], x) is not None and lay._dense_relu(1, lay.hidden_list[1], first.new_zeros(77, 32)) is None
    again = lay(x)
    assert tuple(again.shape) == (77, 8) and float((again - first).abs().max()) < 1e-5


@pytest.mark.parametrize("M,N,K,ta,tb,epi", [(4096, 256, 637, 0, 0, 2), (4096, 637, 256, 0, 1, 0), (637, 256, 4096, 1, 0, 0), (4096, 128, 256, 0, 0, 2),
                                             (256, 128, 4096, 1, 0, 0), (4096, 64, 128, 0, 0, 1), (1, 1, 1, 0, 0, 0), (65, 33, 17, 1, 1, 1),
                                             (130, 70, 1000, 0, 1, 0), (7, 300, 5, 1, 0, 2), (64, 64, 16, 0, 0, 0), (200, 9, 2048, 1, 0, 0),
                                             (1024, 256, 637, 0, 0, 2), (4096, 1, 64, 0, 0, 1), (4096, 64, 1, 0, 1, 0), (64, 1, 4096, 1, 0, 0),
                                             (4096, 2, 144, 0, 0, 1), (144, 2, 4096, 1, 0, 0)])
def test_gemm_f32_matches_float64(M, N, K, ta, tb, epi):
    """fil_gemm_f32 (csrc/gemm.hip: the dense layers' GEMMs -- y = x W + b with ReLU, dx = dz W^T, dW = x^T dz, at the xDeepFM MLP's shapes
    and at ragged ones) against the same product in float64: 1e-5 norm-relative (exact fp32 MFMA chains; split-K shapes sum their slices in
    order), bit-identical on a repeat."""
    from ml_function_amd import functional as Fn
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if epi else None
    c = Fn.gemm_f32(dev(a), dev(b), trans_a=bool(ta), trans_b=bool(tb), bias=dev(bias) if epi else None, relu=epi == 2)
    want = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    if epi:
        want = want + bias
    if epi == 2:
        want = np.maximum(want, 0.0)
    assert tuple(c.shape) == (M, N)
    check("gemm_f32", c, want, tol=1e-5)
    c2 = Fn.gemm_f32(dev(a), dev(b), trans_a=bool(ta), trans_b=bool(tb), bias=dev(bias) if epi else None, relu=epi == 2)
    assert torch.equal(c, c2)
