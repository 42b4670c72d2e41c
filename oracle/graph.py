"""Op-for-op torch-CPU restatement of the reference TF2 graphs (TEST INFRASTRUCTURE).

PARITY UNPINNED -- see oracle/__init__.py.  Every function follows the cited
reference lines literally: one torch op per TF op, same operand order, same
intermediate shapes (the CIN outer product Z is materialised and transposed
twice, exactly as TF would execute it).  dtype follows the inputs: call with
float64 tensors for the truth used in tolerances / gradient checks (autograd),
with float32 tensors for the CPU baseline timed by bench.py.

Paths are relative to /root/reference/kon/model/ctr_model/layer/.
TF/Keras 2.1 semantics relied upon (documented behaviour; TF cannot be run here):
  * Python lists given to tf.matmul are packed on a new leading axis.
  * Conv1D(filters, 1) on channels-last [B, T, C] == per-position x @ kernel[0] + bias.
  * keras Add sums its inputs left to right and broadcasts size-1 dims.
  * K.dot([B,1,D],[D,1]) -> [B,1,1];  K.batch_dot([B,D,1],[B,1,1]) contracts axes (2,1).
  * LayerNormalization(): axis=-1, epsilon=1e-3, biased variance, gamma/beta over the last axis.
"""
import itertools

import torch


# --------------------------------------------------------------------------- glue (A5)
def keras_add(tensors):
    """tf.keras.layers.Add: left-to-right sum with broadcasting of size-1 dims."""
    out = tensors[0]
    for t in tensors[1:]:
        out = out + t
    return out


def stack_layer(inputs, use_flat=True, axis=None):
    """StackLayer.call, core_layer/core_layer.py:49-55 (Flatten each, then Concatenate(axis, default -1))."""
    if use_flat:
        inputs = [t.reshape(t.shape[0], -1) for t in inputs]
    if len(inputs) == 1:
        return inputs[0]
    return torch.cat(inputs, dim=-1 if axis is None else axis)


# --------------------------------------------------------------------------- A1  FM
def inner_layer(inputs, use_add=False):
    """InnerLayer.call (use_inner=True), interactive_layer/interactive_layer.py:59-66.

    inputs: list of F tensors [B,1,K].  Returns the C(F,2) pair products in
    itertools.combinations order, or their keras Add when use_add.
    """
    cross_list = [emb1 * emb2 for emb1, emb2 in itertools.combinations(inputs, 2)]  # :61
    if use_add:
        cross_list = keras_add(cross_list)  # :64-65
    return cross_list


def fm_layer(cross_embed, linear_embed):
    """FmLayer.call with use_add=True, interactive_layer.py:161-170.

    cross_embed: list of F [B,1,K]; linear_embed: list of tensors [B,1,1] (may be empty).
    Returns [B,1,K] = sum_{i<j} e_i*e_j + broadcast sum of the linear terms (no sum over K).
    """
    cross = inner_layer(cross_embed, use_add=True)  # :165 (self.cross built with use_add=True, :153)
    return keras_add([cross] + list(linear_embed))  # :166


# --------------------------------------------------------------------------- A2  DCN cross
def cross_layer(x, kernels, biases):
    """CrossLayer.call, interactive_layer.py:275-282.

    x [B,D]; kernels/biases: lists of cross_hidden tensors [D,1].  Returns [B,D,1] (not squeezed).
    """
    inputs = x.unsqueeze(-1)  # :276  [B,D,1]
    pre_inputs = inputs
    for w, b in zip(kernels, biases):
        s = torch.matmul(pre_inputs.transpose(1, 2), w)  # K.dot([B,1,D],[D,1]) -> [B,1,1]   :279-280
        pre_inputs = torch.bmm(inputs, s) + pre_inputs + b  # K.batch_dot([B,D,1],[B,1,1]) + x_l + bias_l
    return pre_inputs


# --------------------------------------------------------------------------- A3  CIN
def cin(x, conv_kernels, conv_biases, dense_w=None, dense_b=None, output_dim=1):
    """CIN.call, interactive_layer.py:310-327.

    x [B,F,K]; conv_kernels[l] [H_{l-1}*F, H_l] (Keras Conv1D kernel [1,C,H] with the
    leading 1 dropped), conv_biases[l] [H_l]; dense_w [L*K,1], dense_b [1] (Dense(1), :303-304).
    Returns [B,1] if output_dim == 1 else the pooled concat [B, L*K].
    """
    x0 = torch.split(x, 1, dim=-1)  # :311  K tensors [B,F,1]
    pre_ = x0
    sum_pooling_list = []
    for w, b in zip(conv_kernels, conv_biases):
        a = torch.stack(x0)  # list -> packed [K,B,F,1]
        p = torch.stack(pre_)  # [K,B,Hp,1]
        z = torch.matmul(a, p.transpose(-1, -2))  # :316  [K,B,F,Hp]
        z = z.permute(1, 0, 3, 2)  # :317  [B,K,Hp,F]
        z = z.reshape(-1, z.shape[1], z.shape[2] * z.shape[3])  # :318  [B,K,Hp*F], c = h*F+f
        z = torch.matmul(z, w) + b  # :319  Conv1D(size,1): [B,K,H]
        pre_t = z.permute(0, 2, 1)  # :320  [B,H,K]
        pre_ = torch.split(pre_t, 1, dim=-1)  # :321
        sum_pooling_list.append(z.sum(dim=-1))  # :322  sum over feature maps -> [B,K]
    output = torch.cat(sum_pooling_list, dim=-1)  # :323  [B, L*K]
    if output_dim == 1:
        output = torch.matmul(output, dense_w) + dense_b  # :324-325
    return output


def cin_feature_maps(x, conv_kernels, conv_biases):
    """Intermediate x^l [B,H_l,K] of cin() (for kernel-level parity tests)."""
    x0 = torch.split(x, 1, dim=-1)
    pre_ = x0
    maps = []
    for w, b in zip(conv_kernels, conv_biases):
        z = torch.matmul(torch.stack(x0), torch.stack(pre_).transpose(-1, -2)).permute(1, 0, 3, 2)
        z = z.reshape(-1, z.shape[1], z.shape[2] * z.shape[3])
        z = torch.matmul(z, w) + b
        pre_t = z.permute(0, 2, 1)
        pre_ = torch.split(pre_t, 1, dim=-1)
        maps.append(pre_t)
    return maps


# --------------------------------------------------------------------------- A4  AutoInt interacting layer
def product_attention(q, k, v, use_scale=False, mask=None, mask_mod=1):
    """ProductAttentionLayer.call, behavior_layer/behavior_layer.py:292-311 ("softmax" is a sigmoid, :286)."""
    atten_score = torch.matmul(q, k.transpose(-1, -2))  # :294
    if use_scale:
        atten_score = atten_score / (q.shape[-1] ** 0.5)  # :296-297
    if mask is not None:
        if mask_mod == 1:
            atten_score = torch.matmul(atten_score, mask.to(atten_score.dtype))  # :300-302
        if mask_mod == 2:
            atten_score = atten_score + mask.to(atten_score.dtype) * (-100000)  # :303-306
    atten_score = torch.sigmoid(atten_score)  # :308
    return torch.matmul(atten_score, v)  # :309


def layer_norm(x, gamma, beta, eps=1e-3):
    """tf.keras.layers.LayerNormalization() defaults: axis=-1, epsilon=1e-3, biased variance."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def mult_head_attention(x, query_w, key_w, res_w, ln_gamma, ln_beta, use_scale=True, use_res=True,
                        use_ln=True, head_concat=False, mask=None, mask_mod=1):
    """MultHeadAttentionLayer.call, behavior_layer.py:356-377.

    x [B,F,K]; query_w/key_w/res_w [K,H,A].  V is projected with key_w (:360); value_w is unused.
    Returns [atten_v, res], each [H,B,F,A] (atten_v is [B,H,F,A] when head_concat).
    """
    q = torch.tensordot(x, query_w, dims=1).permute(2, 0, 1, 3)  # :358
    k = torch.tensordot(x, key_w, dims=1).permute(2, 0, 1, 3)  # :359
    v = torch.tensordot(x, key_w, dims=1).permute(2, 0, 1, 3)  # :360 (key_w again)
    atten_v = product_attention(q, k, v, use_scale=use_scale, mask=mask, mask_mod=mask_mod)  # :362
    res = []
    if use_res:
        res = torch.tensordot(x, res_w, dims=1).permute(2, 0, 1, 3)  # :365-366
    if use_ln:
        atten_v = layer_norm(atten_v, ln_gamma, ln_beta)  # :368-369
    if head_concat:
        atten_v = atten_v.permute(1, 0, 2, 3)  # :371-372
    if query_w.shape[1] == 1:
        return atten_v.squeeze(0)  # :374-375
    return [atten_v, res]


def autoint_interacting(x, query_w, key_w, res_w, ln_gamma, ln_beta, use_scale=True, use_res=True, use_ln=True):
    """DnnLayer(res_unit=1, other_dense=[MultHeadAttentionLayer]).call, core_layer/core_layer.py:201-226.

    [x, ori] = atten_layer(x); res=[ori, x]; x = Add(res) (skipped when shapes are not
    broadcast-compatible, :211-214); x = ReLU(x).  Returns [H,B,F,A].
    """
    atten_v, ori = mult_head_attention(x, query_w, key_w, res_w, ln_gamma, ln_beta, use_scale=use_scale,
                                       use_res=use_res, use_ln=use_ln)
    if use_res:
        y = keras_add([ori, atten_v])  # res=[ori, x] -> Add, core_layer.py:206-212
    else:
        y = atten_v  # Add([]-containing list) raises ValueError -> x = res[-1], :213-214
    return torch.relu(y)  # ResActivateLayer with bn/ln off, :216


def head_concat(y):
    """[H,B,F,A] -> [B,F,H*A], feature index h*A + a: the reference's own way of feeding multi-head output onward,
    ESULayer.call, behavior_layer.py:973 (tf.split on the head axis, tf.concat on the last axis, tf.squeeze)."""
    return torch.cat(torch.split(y, 1, dim=0), dim=-1).squeeze(0)


def autoint_stack(x, layers, use_scale=True, use_res=True, use_ln=True):
    """BASELINE config 5 ("AutoInt 3-layer"): a stack of interacting layers.  EXTENSION -- the reference builds ONE
    layer (models.py:159-163; a literal second MultHeadAttentionLayer on the [H,B,F,A] output would transpose a rank-5
    tensor at behavior_layer.py:358).  Defined per SURVEY.md section 8 A4: layer l+1 reads head_concat(layer l output).
    layers: list of (query_w, key_w, res_w, ln_gamma, ln_beta).  Returns the last layer's [H,B,F,A]."""
    y = None
    for (qw, kw, rw, gam, bet) in layers:
        y = autoint_interacting(x, qw, kw, rw, gam, bet, use_scale=use_scale, use_res=use_res, use_ln=use_ln)
        x = head_concat(y)
    return y


def autoint_flatten(y):
    """models.py:162 -- [squeeze(h) for h in split(atten_vec, H)] -> StackLayer(use_flat=True, axis=-1): [B, H*F*A]."""
    heads = [h.squeeze(0) for h in torch.split(y, 1, dim=0)]
    return stack_layer(heads, use_flat=True, axis=-1)


# --------------------------------------------------------------------------- N1  SparseEmbed
def sparse_embed(tables, indices):
    """SparseEmbed.call (use_flatten=False, use_add=False), interactive_layer.py:225-242.

    tables: list of F [V_f, K]; indices: list of F integer tensors [B,1].  Returns F x [B,1,K].
    """
    return [t[i.long()] for t, i in zip(tables, indices)]


def linear_layer(inputs, w, b):
    """LinearLayer.call (interactive_layer.py:186-187): tensordot(input, w, axes=1) + b for every input."""
    return [torch.tensordot(t, w, dims=1) + b for t in inputs]


def attention_base_layer(inputs, kernel_w, kernel_b, mlp_kernel, out_w, out_b):
    """AttentionBaseLayer.call (interactive_layer.py:357-364), op for op.  Activation('softmax') is Keras' softmax over
    axis -1; score_ has shape [B,P,1], so the normalisation runs over a single element."""
    x = torch.cat(list(inputs), dim=1)                                   # tf.concat(inputs, axis=1)
    score_ = torch.relu(torch.matmul(torch.matmul(x, kernel_w) + kernel_b, mlp_kernel))   # Dense(1, 'relu', use_bias=False)
    score_w = torch.softmax(score_, dim=-1)
    atten_inputs = torch.sum(score_w * x, dim=1)
    return torch.matmul(atten_inputs, out_w) + out_b


# --------------------------------------------------------------------------- score head and loss (N2)
def score_layer(inputs, use_add=True):
    """ScoreLayer.call, core_layer/core_layer.py:58-84 (use_inner=False, use_global=False): keras Add over the inputs when
    use_add, then tf.sigmoid."""
    x = keras_add(list(inputs)) if use_add else inputs
    return torch.sigmoid(x)


def merge_score_layer(inputs, kernel, bias, use_merge=True):
    """MergeScoreLayer.call, core_layer/core_layer.py:86-100: StackLayer() over the inputs when use_merge (flatten each, concatenate on
    the last axis, :49-55), then tf.keras.layers.Dense(units=output_dim, activation='softmax'): softmax(x @ kernel + bias)."""
    x = stack_layer(list(inputs)) if use_merge else inputs
    return torch.softmax(torch.matmul(x, kernel) + bias, dim=-1)


def binary_crossentropy(y_true, y_pred, eps=1e-7):
    """tf.losses.binary_crossentropy on probabilities, the loss the reference compiles its CTR models with
    (example/ctr_example/un_seq.py:61).  TensorFlow is a third-party dependency absent from /root/reference (README badge:
    tensorflow v2.1); restated from TF 2.1 keras/backend.py binary_crossentropy(from_logits=False):
        output = clip_by_value(output, eps, 1 - eps);  bce = -(target log(output + eps) + (1 - target) log(1 - output + eps))
    then the mean over the batch.  (In graph mode TF 2.1 short-cuts an output that comes straight from a Sigmoid op to
    sigmoid_cross_entropy_with_logits -- the same function of the logits without the clip; the clipped form is what runs on
    probabilities, e.g. the softmax head's column.)  The clip bounds are rounded to the dtype of y_pred's SOURCE (fp32) as TF
    rounds them."""
    lo = float(torch.tensor(eps, dtype=torch.float32))
    hi = float(torch.tensor(1.0, dtype=torch.float32) - torch.tensor(eps, dtype=torch.float32))
    out = torch.clamp(y_pred, lo, hi)
    bce = -(y_true * torch.log(out + eps) + (1 - y_true) * torch.log(1 - out + eps))
    return bce.mean()
