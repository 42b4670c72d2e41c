"""The PRODUCT's field-index front end (ml_function_amd/data_prepare.py) against goldens produced by the real scikit-learn / pandas,
the third-party code the reference calls for this step (/root/reference/kon/utils/data_prepare.py:85-100, :294-301).  Bit-exact."""
import os

import numpy as np
import pandas as pd
import pytest

from ml_function_amd.data_prepare import data_prepare, label_encode, minmax_scale

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _raw_column(tokens):
    """make_golden.label_encode's repr tokens back to the raw Python values."""
    out = []
    for t in tokens:
        t = str(t)
        if t == "n":
            out.append(None)
        elif t.startswith("f:"):
            out.append(float(t[2:]))
        elif t.startswith("i:"):
            out.append(int(t[2:]))
        elif t.startswith("b:"):
            out.append(bool(int(t[2:])))
        else:
            out.append(t[2:])
    return np.asarray(out, dtype=object)


def _frame():
    g = np.load(os.path.join(GOLD, "label_encode.npz"))
    names = [str(n) for n in g["names"]]
    return g, names, pd.DataFrame({n: _raw_column(g["raw_" + n]) for n in names})


def test_label_encode_ids_match_sklearn_bit_for_bit():
    g, names, df = _frame()
    for f, n in enumerate(names):
        ids, classes = label_encode(df[n])
        assert ids.dtype == np.int64 and np.array_equal(ids, g["idx"][:, f]), n
        assert len(classes) == int(g["vocab"][f]), n
        # the string form the reference's fillna('-1').astype('str') produces, and its order ('10' < '2', 'B' < 'a', '-1' first)
        assert np.array_equal(classes[ids], g["col_" + n]), n
    c1 = [str(c) for c in label_encode(df["C1"])[1]]
    assert c1 == sorted(c1) and c1.index("10") < c1.index("2")


def test_sparse_fea_deal_descriptors_and_ids():
    g, names, df = _frame()
    before = df.copy()
    ids, info = data_prepare(batch_size=32).sparse_fea_deal(df, embed_dim=16, linear_dim=1)
    assert df.equals(before)                                              # the caller's frame is untouched
    assert list(ids.columns) == names and np.array_equal(ids.to_numpy(), g["idx"])
    for f, fea in enumerate(info):
        assert fea.fea_name == names[f] and fea.word_size == int(g["vocab"][f]) and fea.input_dim == len(df)
        assert (fea.cross_unit, fea.linear_unit, fea.input_length, fea.batch_size) == (16, 1, 1, 32)
        assert fea.pre_weight is None and fea.is_trainable is True and fea.mask_zero is False and fea.emb_reg == 1e-8
    _, info2 = data_prepare().sparse_fea_deal(df, pre_weight=list("abcde"), emb_reg=[0.1 * i for i in range(1, 6)])
    assert [f.pre_weight for f in info2] == list("abcde") and info2[2].emb_reg == pytest.approx(0.3)


def test_dense_fea_deal_matches_sklearn_minmax_bit_for_bit():
    g = np.load(os.path.join(GOLD, "dense_minmax.npz"))
    names = [str(n) for n in g["names"]]
    df = pd.DataFrame(g["raw"], columns=names)
    out, info = data_prepare(batch_size=8).dense_fea_deal(df)
    assert np.array_equal(out.to_numpy(), g["out"])                       # same float64 bits as MinMaxScaler
    assert [d.fea_name for d in info] == names and info[0].batch_size == 8
    assert out.to_numpy().min() == 0.0 and out.to_numpy().max() == 1.0
    assert np.all(out["I3"].to_numpy() == 0.0)                            # constant column: (x - min) * 1
    assert np.isnan(df["I4"]).any() and not np.isnan(out["I4"]).any()     # missing values took the mode
    assert np.array_equal(minmax_scale(pd.DataFrame({f: df[f].fillna(df[f].mode()[0]) for f in df}).to_numpy()), g["out"])


def test_concat_and_tensors():
    dp = data_prepare()
    a = pd.DataFrame({"c": ["x", "y", None]})
    b = pd.DataFrame({"c": ["y", "z"]})
    df, (tr, te) = dp.concat_test_train(a, b)
    assert tr == [0, 1, 2] and te == [3, 4] and len(df) == 5
    ids, info = dp.sparse_fea_deal(df)
    assert ids["c"].tolist() == [1, 2, 0, 2, 3] and info[0].word_size == 4      # '-1' < 'x' < 'y' < 'z'
    dense, sparse = dp.to_tensors(None, ids)
    assert dense is None and sparse.dtype.is_floating_point is False and tuple(sparse.shape) == (5, 1)


@pytest.mark.gpu
def test_frame_to_ids_to_gather_bit_exact():
    """frame -> product front end -> ids -> fil_embed_gather: the rows the kernels return are bit-identical copies of the table rows the
    scikit-learn ids select (tests/golden/label_encode.npz: `gathered` was built from sklearn's ids on the CPU)."""
    import torch
    from ml_function_amd import functional as Fn
    g, names, df = _frame()
    ids, info = data_prepare().sparse_fea_deal(df, embed_dim=8)
    vocab = [f.word_size for f in info]
    assert vocab == [int(v) for v in g["vocab"]]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), dtype=torch.int64, device="cuda")
    sizes = torch.tensor(vocab, dtype=torch.int64, device="cuda")
    table = torch.tensor(g["table"], device="cuda")
    out = Fn.embed_gather(table, offsets, torch.tensor(ids.to_numpy(), device="cuda"), sizes=sizes)
    assert np.array_equal(out.cpu().numpy(), g["gathered"])
