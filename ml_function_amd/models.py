"""Model assembly around the interaction layers: the re-host of the non-sequence part of the reference's model zoo
(kon/model/ctr_model/model/models.py: FM :36-41, DeepFM :80-90, DCN :92-106, XDeepFM :121-138, AutoInt :150-165) and of
the feature-input plumbing it relies on (kon/utils/data_prepare.py: InputFeature :39-54, sparseFea/denseFea :59-60,
FeatureInput :65-76).  The reference builds symbolic Keras models; here every zoo entry is a Layer whose call takes
the concrete InputFeature of one batch, and ``CTRModel`` chains FeatureInput -> zoo body on raw (dense, sparse-id)
tensors.  Hyper-parameter names and defaults are the reference's.  Sequence models are out of scope.
"""
from collections import namedtuple

import torch

from .layers import (CIN, AttentionBaseLayer, CrossLayer, DnnLayer, FmLayer, InnerLayer, IPnnLayer, MergeScoreLayer,
                     MultHeadAttentionLayer, OPnnLayer, ScoreLayer, SparseEmbed, StackLayer)
from .layers.base import Layer
from .layers.interactive_layer import pack_fields
from .layers.core_layer import keras_add

sparseFea = namedtuple("sparseFea", ["fea_name", "word_size", "input_dim", "cross_unit", "linear_unit", "pre_weight", "mask_zero",
                                     "is_trainable", "input_length", "sample_num", "batch_size", "emb_reg"])
denseFea = namedtuple("denseFea", ["fea_name", "batch_size"])


def make_sparse_info(word_sizes, embed_dim=8, linear_dim=1, names=None, batch_size=None):
    """sparseFea descriptors as data_prepare.sparse_fea_deal builds them (data_prepare.py:95-100), from vocabulary sizes."""
    names = names or ["C%d" % (i + 1) for i in range(len(word_sizes))]
    return [sparseFea(fea_name=n, word_size=int(v), input_dim=None, cross_unit=embed_dim, linear_unit=linear_dim, pre_weight=None,
                      mask_zero=False, is_trainable=True, input_length=1, sample_num=None, batch_size=batch_size, emb_reg=1e-8)
            for n, v in zip(names, word_sizes)]


class InputFeature(object):
    """Same attribute names as the reference's InputFeature (data_prepare.py:39-54)."""

    def __init__(self, denseInfo=None, sparseInfo=None, seqInfo=None, denseInputs=None, sparseInputs=None, seqInputs=None,
                 linearEmbed=None, sparseEmbed=None, seqEmbedList=None):
        self.dense_info = denseInfo
        self.sparse_info = sparseInfo
        self.seq_info = seqInfo
        self.dense_inputs = denseInputs
        self.sparse_inputs = sparseInputs
        self.seq_inputs = seqInputs
        self.linear_embed = linearEmbed
        self.sparse_embed = sparseEmbed
        self.seq_embed_list = seqEmbedList


class FeatureInput(Layer):
    """data_prepare.FeatureInput (data_prepare.py:65-76): embeds the sparse ids (cross embeddings [B,1,K] per field,
    optional linear embeddings) and passes the dense columns through as F_d tensors [B,1]."""

    def __init__(self, sparseInfo=None, denseInfo=None, useLinear=False, useAddLinear=False, useFlattenLinear=False,
                 useFlattenSparse=False, emitXT=False, embedDtype=None):
        super().__init__()
        self.sparse_info = sparseInfo or []
        self.dense_info = denseInfo or []
        self.use_linear = useLinear
        # emitXT (extension): the embedding gather also writes the block in the layout the CIN kernels read (XDeepFM)
        # embedDtype (extension): torch.bfloat16 = the cross embeddings leave the gather as bf16 (a bf16 model's cast of the block, fused)
        self.sparse_embed = (SparseEmbed(self.sparse_info, use_flatten=useFlattenSparse, emit_xt=emitXT, out_dtype=embedDtype)
                             if self.sparse_info else None)
        self.linear_embed = (SparseEmbed(self.sparse_info, use_flatten=useFlattenLinear, is_linear=True, use_add=useAddLinear)
                             if (useLinear and self.sparse_info) else None)

    def call(self, inputs, **kwargs):
        dense, sparse_idx = inputs
        dense_inputs = [] if dense is None else [dense[:, i:i + 1] for i in range(dense.shape[1])]
        sparse_inputs = [] if sparse_idx is None else [sparse_idx[:, i:i + 1] for i in range(sparse_idx.shape[1])]
        linear = self.linear_embed(sparse_idx) if self.linear_embed is not None else None
        embed = self.sparse_embed(sparse_idx) if self.sparse_embed is not None else None
        return InputFeature(self.dense_info, self.sparse_info, None, dense_inputs, sparse_inputs, [], linear, embed, [None, None])


class FM(torch.nn.Module):
    """models.FM (:36-41): FmLayer -> squeeze -> MergeScoreLayer(use_merge=False)."""

    def __init__(self):
        super().__init__()
        self.fm = FmLayer()
        self.score = MergeScoreLayer(use_merge=False)

    def forward(self, inputFea):
        fm_ = self.fm([inputFea.sparse_embed, inputFea.linear_embed])
        return self.score(fm_.squeeze(1))


class PNN(torch.nn.Module):
    """models.PNN (:42-55).  use_outer=True goes through OPnnLayer, which raises AttributeError in the reference
    (and here); the default constructor keeps the reference's defaults, so pass use_outer=False to run it.
    linear_embed must be a list (FeatureInput(useLinear=True))."""

    def __init__(self, hidden_units=None, use_inner=True, use_outer=True):
        super().__init__()
        self.use_inner, self.use_outer = use_inner, use_outer
        self.ipnn, self.opnn = IPnnLayer(), OPnnLayer()
        self.stack = StackLayer()
        self.dnn = DnnLayer(hidden_units or [256, 256, 256])
        self.score = MergeScoreLayer(use_merge=False)

    def forward(self, inputFea):
        cross_fea = list(inputFea.linear_embed)
        if self.use_inner:
            cross_fea += self.ipnn(inputFea.sparse_embed)
        if self.use_outer:
            cross_fea += self.opnn(inputFea.sparse_embed)
        return self.score(self.dnn(self.stack(cross_fea)))


class DeepCross(torch.nn.Module):
    """models.DeepCross (:57-66): the body only runs when hidden_units is None (its indentation in the reference puts
    everything under that `if`); with explicit hidden_units the reference returns None, and so does this."""

    def __init__(self, hidden_units=None):
        super().__init__()
        self.given = hidden_units is not None
        self.stack = StackLayer()
        self.dnn = DnnLayer(hidden_units=[256, 256, 256])
        self.score = MergeScoreLayer(use_merge=False)

    def forward(self, inputFea):
        if self.given:
            return None
        return self.score(self.dnn(self.stack(list(inputFea.dense_inputs) + list(inputFea.sparse_embed))))


class Wide_Deep(torch.nn.Module):
    """models.Wide_Deep (:68-78)."""

    def __init__(self, hidden_units=None):
        super().__init__()
        self.stack = StackLayer()
        self.dnn = DnnLayer(hidden_units=hidden_units or [256, 128, 64])
        self.score = MergeScoreLayer()

    def forward(self, inputFea):
        dnn_ = self.dnn(self.stack(list(inputFea.dense_inputs) + list(inputFea.sparse_embed)))
        return self.score(list(inputFea.linear_embed) + [dnn_])


class NFM(torch.nn.Module):
    """models.NFM (:108-119): bi-interaction (sum of pair products) -> DNN(output_dim=1) + linear terms -> sigmoid."""

    def __init__(self, hidden_units=None):
        super().__init__()
        self.inner = InnerLayer(use_inner=True, use_add=True)
        self.stack = StackLayer()
        self.dnn = DnnLayer(hidden_units=hidden_units or [256, 128, 64], output_dim=1)
        self.score = ScoreLayer()

    def forward(self, inputFea):
        cross_inputs = self.inner(inputFea.sparse_embed)
        dnn_fea = self.dnn(self.stack(list(inputFea.dense_inputs) + [cross_inputs]))
        return self.score(keras_add(list(inputFea.linear_embed) + [dnn_fea]))


class AFM(torch.nn.Module):
    """models.AFM (:141-147)."""

    def __init__(self):
        super().__init__()
        self.inner = InnerLayer()
        self.atten = AttentionBaseLayer()
        self.score = ScoreLayer(use_add=True)

    def forward(self, inputFea):
        atten_output = self.atten(self.inner(inputFea.sparse_embed))
        return self.score(list(inputFea.linear_embed) + [atten_output])


class DeepFM(torch.nn.Module):
    """models.DeepFM (:80-90)."""

    def __init__(self, hidden_units=None):
        super().__init__()
        self.fm = FmLayer()
        self.stack = StackLayer()
        self.dnn = DnnLayer(hidden_units=hidden_units or [256, 128, 64])
        self.score = MergeScoreLayer()

    def forward(self, inputFea):
        fm_ = self.fm([inputFea.sparse_embed, inputFea.linear_embed])
        dnn_ = self.dnn(self.stack(list(inputFea.dense_inputs) + list(inputFea.sparse_embed)))
        return self.score([fm_, dnn_])


class DCN(torch.nn.Module):
    """models.DCN (:92-106)."""

    def __init__(self, hidden_units=None, cross_hidden=3):
        super().__init__()
        self.stack = StackLayer()
        self.cross = CrossLayer(cross_hidden=cross_hidden)
        self.deep = DnnLayer(hidden_units=hidden_units or [256, 128, 64])
        self.score = MergeScoreLayer()

    def forward(self, inputFea):
        combine_inputs = self.stack(list(inputFea.dense_inputs) + list(inputFea.sparse_embed))
        return self.score([self.cross(combine_inputs), self.deep(combine_inputs)])


class XDeepFM(torch.nn.Module):
    """models.XDeepFM (:121-138).  linear_embed must be one tensor (FeatureInput(useLinear, useAddLinear, useFlattenLinear))."""

    def __init__(self, conv_size=None, hidden_units=None):
        super().__init__()
        self.stack = StackLayer()
        self.cin = CIN(conv_size=conv_size or [200, 200, 200], output_dim=1)
        self.dnn = DnnLayer(hidden_units=hidden_units or [256, 128, 64], output_dim=1)
        self.score = ScoreLayer(use_add=True)

    def forward(self, inputFea):
        cin_inputs = pack_fields(list(inputFea.sparse_embed))    # Concatenate(axis=1); views of one packed block re-pack without a copy
        dnn_inputs = self.stack(list(inputFea.dense_inputs) + list(inputFea.sparse_embed))
        return self.score([inputFea.linear_embed, self.cin(cin_inputs), self.dnn(dnn_inputs)])


class AutoInt(torch.nn.Module):
    """models.AutoInt (:150-165)."""

    def __init__(self, attention_dim=8, attention_head_dim=3):
        super().__init__()
        self.stack_embed = StackLayer(use_flat=False, axis=1)
        self.atten_layer = MultHeadAttentionLayer(attention_dim=attention_dim, attention_head_dim=attention_head_dim, use_ln=True,
                                                  atten_mask_mod=1)
        self.dnn = DnnLayer(res_unit=1, other_dense=[self.atten_layer])
        self.stack_flat = StackLayer(use_flat=True, axis=-1)
        self.score = MergeScoreLayer(use_merge=False)

    def forward(self, inputFea):
        cross_embed = self.stack_embed(list(inputFea.sparse_embed))
        atten_vec = self.dnn(cross_embed)
        final_input = self.stack_flat([h.squeeze(0) for h in torch.split(atten_vec, 1, dim=0)])
        return self.score(final_input)


class CTRModel(torch.nn.Module):
    """FeatureInput + zoo body on raw batch tensors: model(dense [B, n_dense] or None, sparse_ids [B, F])."""

    def __init__(self, feature_input, body):
        super().__init__()
        self.feature_input = feature_input
        self.body = body

    def forward(self, dense, sparse_idx):
        return self.body(self.feature_input((dense, sparse_idx)))
