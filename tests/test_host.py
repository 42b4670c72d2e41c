"""CPU-side checks: the C-ABI library loads and exports every symbol of include/fil.h, argument validation and
workspace sizing work without a GPU, the layer classes keep the reference's constructor surface / weight names /
error behaviour, and the product path refuses to run on the CPU (no fallback)."""
import ctypes
import os
import inspect

import numpy as np
import subprocess
import sys

import pytest
import torch

from ml_function_amd import _lib, layers
from ml_function_amd._lib import FilError


@pytest.fixture(scope="module")
def lib():
    from ml_function_amd import build
    build.build(verbose=False)
    return _lib.load()


def test_library_exports_every_header_symbol(lib):
    syms = _lib.header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libfil_hip.so does not export %s" % s
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert lib.fil_version() == _lib.header_abi_version() >= 200   # a stale .so is refused by _lib.load()


def test_argument_validation_without_gpu(lib):
    assert lib.fil_fm_fwd(None, None, None, 4, 3, 2, 0, None) == -1
    assert b"bad argument" in lib.fil_last_error()
    assert lib.fil_fm_fwd(None, None, None, 0, 3, 2, 0, None) == 0  # empty batch is a no-op
    assert lib.fil_fm_fwd(None, None, None, 4, 3, 2, 7, None) == -1  # unknown dtype
    assert lib.fil_dcn_fwd(None, None, None, None, None, 4, 5000, 17, None) == -4  # L limit (any D runs since round 6)
    assert b"16" in lib.fil_last_error()
    H = _lib.int_array([128, 128, 128])
    assert lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 9, H, 1, 0, None, 0, None) == -4
    Hbig = _lib.int_array([300])
    assert lib.fil_cin_fwd(None, None, None, None, None, None, None, None, 4, 39, 16, 1, Hbig, 1, 0, None, 0, None) == -4
    assert lib.fil_attn_fwd(None, None, None, None, None, None, None, None, None, None, 4, 200, 16, 4, 32, 0.25, 1e-3, 1, 0, 0, None, 0, None) == -4
    # unknown precision code
    assert lib.fil_attn_fwd(None, None, None, None, None, None, None, None, None, None, 4, 200, 16, 4, 16, 0.25, 1e-3, 1, 7, 0, None, 0, None) == -1
    # head-major input: the chunk width must divide K; more than 8 heads is outside the one-wave-per-head design
    assert lib.fil_attn_fwd(None, None, None, None, None, None, None, None, None, None, 4, 200, 64, 4, 16, 0.25, 1e-3, 1, 0, 24, None, 0, None) == -1
    assert lib.fil_attn_fwd(None, None, None, None, None, None, None, None, None, None, 4, 200, 16, 9, 16, 0.25, 1e-3, 1, 0, 0, None, 0, None) == -4


def test_every_entry_point_validates_its_arguments(lib):
    from tests import host_calls
    assert host_calls.run(lib) >= 30


def test_host_shim_under_asan_ubsan():
    """The same calls against the AddressSanitizer + UBSan build of the host shim (hipcc instruments the host pass only: GPU
    sanitizers are not available on this pool).  A heap / stack overrun, use-after-free or undefined behaviour in argument
    checking, workspace sizing, launch planning (dw_plan, bwd_grid ...) or error formatting aborts the child."""
    from ml_function_amd import build as _build
    asan_lib = _build.build_asan()
    rt = _build.asan_runtime()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "host_calls.py"), asan_lib], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "host calls ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]


def test_product_attention_entry_points_validate(lib):
    assert lib.fil_pattn_fwd(None, None, None, None, None, 4, 39, 39, 80, 8, 1.0, 0, None) == -4      # A > 64
    assert lib.fil_pattn_fwd(None, None, None, None, None, 0, 39, 39, 8, 8, 1.0, 0, None) == 0       # empty
    assert lib.fil_pattn_fwd(None, None, None, None, None, 4, 39, 39, 8, 8, 1.0, 0, None) == -1      # NULL tensors
    assert lib.fil_pattn_bwd(None, None, None, None, None, None, None, None, 4, 0, 39, 8, 8, 1.0, 0, None) == -1


def test_embedding_entry_points_validate(lib):
    assert lib.fil_embed_gather(None, None, None, None, None, None, 4, 3, 8, None) == -1
    assert lib.fil_embed_gather(None, None, None, None, None, None, 0, 3, 8, None) == 0
    assert lib.fil_embed_row_ids(None, None, None, None, None, 4, 0, None) == -1
    assert lib.fil_embed_segment_sum(None, None, None, None, None, None, 5, 300, None) == -4
    assert lib.fil_embed_run_sum(None, None, None, None, 0, 8, None) == 0


def test_workspace_sizes(lib):
    H = _lib.int_array([128, 128, 128])
    B, F, K = 4096, 39, 16
    M = B * K
    # saved = xT [M][F] + the first layer's m-major feature map [M][128] + the quadratic tail's R [M][128], T [F*F][128], wsum_L [128*F]
    # and cvec [128] (one 256-byte-aligned slice); the upper layers' maps are never stored
    # ... + wsum_p [128*F], its MFMA operand copy [40*128] and T in the dZ kernel's slot order [26 tiles * 32 * 128]
    qt = (M * 128 + F * F * 128 + 128 * F + 128 + 128 * F + 40 * 128 + 26 * 32 * 128) * 4
    assert lib.fil_cin_saved_bytes(B, F, K, 3, H) == M * F * 4 + M * 128 * 4 + (qt + 255) // 256 * 256
    assert lib.fil_cin_fwd_workspace_bytes(B, F, K, 3, H) >= 3 * M * 4
    assert lib.fil_cin_bwd_workspace_bytes(B, F, K, 3, H) > 2 * M * 128 * 4
    assert lib.fil_cin_saved_bytes(B, F, K, 1, H) == M * F * 4
    assert lib.fil_dcn_bwd_workspace_bytes(8192, 1248, 3) > 0
    # saved av / y: only the per-workgroup partial sums; otherwise room to re-run the forward (av and y, [H,B,F,A] each)
    small = lib.fil_attn_bwd_workspace_bytes(16, 200, 16, 4, 16, 1)
    assert 0 < small < 1 << 20
    assert lib.fil_attn_bwd_workspace_bytes(16, 200, 16, 4, 16, 0) >= small + 2 * 4 * 16 * 200 * 16 * 4
    assert lib.fil_cin_bwd_workspace_bytes(0, F, K, 3, H) >= 0


def test_constructor_surface_matches_reference():
    sig = lambda c: {k: v.default for k, v in inspect.signature(c.__init__).parameters.items() if k not in ("self", "kwargs")}
    assert sig(layers.InnerLayer) == dict(use_inner=True, mod=1, seed=2020, perm=None, use_add=False)
    assert sig(layers.FmLayer) == dict(use_inner=True, mod=1, use_add=True)
    assert sig(layers.CrossLayer) == dict(cross_hidden=3, seed=2020)
    s = sig(layers.CIN)
    assert s["conv_size"] is None and s["output_dim"] == 1
    assert layers.CIN().conv_size == [200, 200, 200]
    m = sig(layers.MultHeadAttentionLayer)
    assert m["seed"] == 2020 and m["use_scale"] is True and m["use_res"] is True and m["use_ln"] is True
    assert m["head_concat"] is False and m["atten_mask_mod"] == 1
    assert sig(layers.ProductAttentionLayer) == dict(use_scale=False, supports_masking=True, mask_mod=1)
    d = sig(layers.DnnLayer)
    assert d["res_unit"] == 1 and d["output_dim"] == -1 and d["use_bn"] is False and d["other_dense"] is None


def test_build_creates_reference_weights():
    cross = layers.CrossLayer(cross_hidden=3)
    cross.build((None, 1248))
    names = dict(cross.named_parameters())
    assert sorted(names) == ["outer_bias_%d" % i for i in range(3)] + ["outer_weight_%d" % i for i in range(3)]
    assert names["outer_weight_0"].shape == (1248, 1) and float(names["outer_bias_1"].abs().sum()) == 0.0
    lim = np.sqrt(6.0 / (1248 + 1))
    assert float(names["outer_weight_0"].abs().max()) <= lim
    # same seed -> the reference gives every cross layer the same initial kernel (glorot_uniform(seed=self.seed))
    assert torch.equal(names["outer_weight_0"], names["outer_weight_2"])

    cin = layers.CIN(conv_size=[128, 128, 128])
    cin.build((None, 39, 16))
    p = dict(cin.named_parameters())
    assert p["hidden_conv_0_kernel"].shape == (1, 39 * 39, 128)
    assert p["hidden_conv_1_kernel"].shape == (1, 128 * 39, 128)
    assert p["hidden_conv_2_bias"].shape == (128,)
    assert p["logit_layer_kernel"].shape == (48, 1) and p["logit_layer_bias"].shape == (1,)
    assert sum(v.numel() for v in p.values()) == 1473073  # the all-reduce bucket

    att = layers.MultHeadAttentionLayer(attention_dim=16, attention_head_dim=4)
    att.build((None, 200, 16))
    p = dict(att.named_parameters())
    for n in ("query_w", "key_w", "value_w", "res_w"):
        assert p[n].shape == (16, 4, 16)
    assert torch.equal(p["query_w"], p["key_w"])  # same seed, same shape -> same init, like glorot_uniform(seed)
    assert p["ln_gamma"].shape == (16,) and float(p["ln_gamma"].sum()) == 16.0


def test_error_behaviour():
    with pytest.raises(AttributeError):
        layers.InnerLayer(use_inner=False)([torch.zeros(2, 1, 4)] * 3)
    from ml_function_amd.layers.core_layer import keras_add
    with pytest.raises(ValueError):
        keras_add([torch.zeros(2, 3), []])
    with pytest.raises(ValueError):
        keras_add([torch.zeros(2, 3), torch.zeros(2, 4)])
    assert keras_add([torch.ones(2, 1, 4), torch.ones(2, 1, 1)]).shape == (2, 1, 4)


def test_no_cpu_fallback():
    x = torch.zeros(4, 39, 16)
    with pytest.raises(FilError):
        layers.CIN([8, 8])(x)
    with pytest.raises(FilError):
        layers.FmLayer()([[x[:, :1]] * 3, [x[:, :1, :1]] * 3])
    with pytest.raises(FilError):
        layers.CrossLayer()(torch.zeros(4, 20))
    with pytest.raises(FilError):
        layers.DnnLayer(res_unit=1, other_dense=[layers.MultHeadAttentionLayer(8, 3)])(x)


def test_dnn_layer_mlp_path_and_residual_rule():
    torch.manual_seed(0)
    dnn = layers.DnnLayer(hidden_units=[8, 8, 4], output_dim=1)
    x = torch.randn(5, 8)
    y = dnn(x)
    assert y.shape == (5, 1)
    # layer 0: in 8 -> 8, residual added; layer 2: 8 -> 4, shapes differ -> residual skipped (reference :211-214)
    h0 = dnn.hidden_list[0].dense
    want0 = torch.relu(x + (x @ h0.kernel + h0.bias))
    h1 = dnn.hidden_list[1].dense
    want1 = torch.relu(want0 + (want0 @ h1.kernel + h1.bias))
    h2 = dnn.hidden_list[2].dense
    want2 = torch.relu(want1 @ h2.kernel + h2.bias)
    want = want2 @ dnn.logit_layer.kernel + dnn.logit_layer.bias
    assert torch.allclose(y, want, atol=1e-6)


def test_stack_and_score_layers():
    a, b = torch.ones(2, 1, 3), 2 * torch.ones(2, 1, 3)
    assert layers.StackLayer()([a, b]).shape == (2, 6)
    assert layers.StackLayer(use_flat=False, axis=1)([a, b]).shape == (2, 2, 3)
    s = layers.ScoreLayer(use_add=True)([a[:, 0, :1], b[:, 0, :1]])
    assert torch.allclose(s, torch.sigmoid(torch.full((2, 1), 3.0)))
    m = layers.MergeScoreLayer(use_merge=False)(torch.zeros(4, 5))
    assert m.shape == (4, 2) and torch.allclose(m.sum(-1), torch.ones(4))


def test_model_zoo_constructor_surface():
    """models.py zoo defaults follow the reference (models.py:80-165): DNN [256,128,64], CIN [200,200,200], 3 cross layers,
    AutoInt attention_dim=8 x 3 heads with layer norm and the relu-fused residual path."""
    from ml_function_amd import models
    info = models.make_sparse_info([5, 7, 9], embed_dim=4)
    assert info[0]._fields == ("fea_name", "word_size", "input_dim", "cross_unit", "linear_unit", "pre_weight", "mask_zero",
                               "is_trainable", "input_length", "sample_num", "batch_size", "emb_reg")
    fi = models.FeatureInput(sparseInfo=info, useLinear=True, useAddLinear=True, useFlattenLinear=True)
    assert fi.linear_embed.use_add and fi.linear_embed.is_linear and not fi.sparse_embed.use_flatten
    x = models.XDeepFM()
    assert x.cin.conv_size == [200, 200, 200] and x.cin.output_dim == 1
    assert models.DCN().cross.cross_hidden == 3
    a = models.AutoInt()
    assert a.atten_layer.attention_dim == 8 and a.atten_layer.attention_head_dim == 3 and a.atten_layer.use_ln
    for cls in (models.FM, models.DeepFM, models.DCN, models.XDeepFM, models.AutoInt):
        assert isinstance(cls(), __import__("torch").nn.Module)


def test_auc_matches_sklearn():
    import numpy as np
    import torch
    from sklearn.metrics import roc_auc_score
    from ml_function_amd import metrics
    rng = np.random.default_rng(0)
    y = rng.integers(0, 2, 500)
    s = np.round(rng.random(500), 2)  # plenty of ties
    assert abs(metrics.auc(torch.tensor(y), torch.tensor(s)) - roc_auc_score(y, s)) < 1e-12
    assert metrics.auc(torch.tensor([0, 0, 1, 1]), torch.tensor([0.1, 0.2, 0.8, 0.9])) == 1.0
    import pytest
    with pytest.raises(ValueError):
        metrics.auc(torch.tensor([1, 1]), torch.tensor([0.3, 0.4]))


def test_pmc_traffic_summary_and_bench_lookup(tmp_path, monkeypatch):
    """tools/pmc_traffic.py: (2*FETCH_SIZE + WRITE_SIZE)*1024 per kernel and launch slot; bench.pmc_traffic maps the
    profiler scope of the dominant kernel onto it."""
    import json
    import os
    import subprocess
    import sys
    hdr = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
           '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
           '"Start_Timestamp","End_Timestamp"\n')

    def rows(counter, vals):
        out, d = hdr, 0
        for it in range(2):  # two step iterations: fwd l1, fwd l2, dz l2, dz l1
            for name, grid, v in vals:
                d += 1
                out += '%d,%d,"Agent 2",1,1,1,%d,6,"%s",256,0,0,8,0,32,"%s",%f,0,1\n' % (d, d, grid, name, counter, v)
        return out

    k_f1 = "void fil::cin_fwd3_kernel<2, 10, true, 1>(float const*)"
    k_f2 = "void fil::cin_fwd3_kernel<2, 20, false, 1>(float const*)"
    k_z2 = "void fil::cin_dz3_kernel<1, 20, 64, false, 1>(float const*)"
    k_z1 = "void fil::cin_dz3_kernel<2, 10, 64, true, 1>(float const*)"
    (tmp_path / "f").mkdir()
    (tmp_path / "w").mkdir()
    (tmp_path / "f" / "1_counter_collection.csv").write_text(rows("FETCH_SIZE", [(k_f1, 64, 10.0), (k_f2, 64, 20.0), (k_z2, 128, 30.0), (k_z1, 64, 40.0)]))
    (tmp_path / "w" / "1_counter_collection.csv").write_text(rows("WRITE_SIZE", [(k_f1, 64, 1.0), (k_f2, 64, 2.0), (k_z2, 128, 3.0), (k_z1, 64, 4.0)]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = tmp_path / "profiles"
    prof.mkdir()
    out = prof / "r99_pmc_traffic.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(tmp_path / "f"), str(tmp_path / "w"), str(out), "2"],
                   check=True, capture_output=True)
    per = json.loads(out.read_text())["per_launch"]
    assert per["cin_fwd3_kernel<2,20,false,1> grid=64"]["hbm_bytes"] == (2 * 20.0 + 2.0) * 1024
    import bench
    monkeypatch.setattr(bench.os.path, "dirname", lambda p: str(tmp_path))  # bench looks under <its dir>/profiles
    # (a lookup of a committed file, labelled as such in the JSON line)
    assert bench.pmc_traffic("cin_fwd_l2") == ((2 * 20.0 + 2.0) * 1024, "committed profile r99_pmc_traffic.json")
    assert bench.pmc_traffic("cin_fwd_l1")[0] == (2 * 10.0 + 1.0) * 1024
    assert bench.pmc_traffic("cin_bwd_dz_l2")[0] == (2 * 30.0 + 3.0) * 1024   # backward visits layer 2 first
    assert bench.pmc_traffic("cin_bwd_dz_l1")[0] == (2 * 40.0 + 4.0) * 1024
    assert bench.pmc_traffic("cin_head_fwd") == (None, None)


def test_extract_pool_align_layers():
    """The list / pooling glue of SURVEY section 2 rows 1-2 (interactive_layer.py:82-109, core_layer.py:228-258): pure torch, CPU."""
    from collections import namedtuple
    from ml_function_amd.layers import AlignLayer, ExtractLayer, IntraViewPoolingLayer
    Inp = namedtuple("Inp", ["name"])
    descr = [Inp("C1:0"), Inp("C2:0"), "I3", Inp("C4:0")]
    ts = [torch.full((2, 1, 3), float(i)) for i in range(4)]
    ex = ExtractLayer(need_fea=["C2", "I3"], need_inputs=descr)
    out = ex(ts)
    assert [float(t[0, 0, 0]) for t in out] == [1.0, 2.0] and ex.need_idx == [1, 2]
    assert ex.compute_mask(ts, mask=["m0", "m1", "m2", "m3"]) is None
    ex2 = ExtractLayer(need_fea=["C4"], need_inputs=descr, mask_zero=True, need_remove=True)
    picked, rest = ex2(ts)
    assert [float(t[0, 0, 0]) for t in picked] == [3.0] and [float(t[0, 0, 0]) for t in rest] == [0.0, 1.0, 2.0]
    assert ex2.compute_mask(ts, mask=["m0", "m1", "m2", "m3"]) == ["m3"]
    x = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    p = IntraViewPoolingLayer()(x)
    assert p.shape == (2, 1, 4) and torch.allclose(p[:, 0], x.mean(1))
    al = AlignLayer()
    ins = [torch.randn(5, 4), torch.randn(5, 7), torch.randn(5, 2, 7), torch.randn(5, 3)]
    outs = al(ins)
    assert [o.shape[-1] for o in outs] == [7, 7, 7, 7] and outs[1] is ins[1] and outs[2] is ins[2]
    assert [fd is None for fd in al.format_dense] == [False, True, True, False]
    assert torch.allclose(outs[0], ins[0] @ al.format_dense[0].kernel + al.format_dense[0].bias)
    assert float(al.format_dense[3].bias.abs().sum()) == 0.0 and len(list(al.parameters())) == 4
