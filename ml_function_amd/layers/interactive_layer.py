"""Feature-interaction layers: the MI355X re-host of the reference's
kon/model/ctr_model/layer/interactive_layer/interactive_layer.py (InnerLayer :34, FmLayer :145, SparseEmbed :189,
CrossLayer :250, CIN :285).  Same constructor kwargs, weight names, call signatures and output shapes; the
per-batch arithmetic runs in the HIP kernels behind include/fil.h (no CPU fallback).

Packed [B,F,K] is the native layout; the reference's Python lists of F tensors [B,1,K] are accepted and stacked.
"""
import os

import torch

from .. import functional as Fn
from .base import Layer, glorot_uniform_, merge_packed_views


def pack_fields(inputs):
    """list of F tensors [B,1,K] (or [B,K]) -> [B,F,K]; a packed tensor passes through."""
    if isinstance(inputs, (list, tuple)):
        ts = merge_packed_views([t if t.dim() == 3 else t.unsqueeze(1) for t in inputs])
        return ts[0] if len(ts) == 1 else torch.cat(ts, dim=1)
    return inputs


def pack_linear(inputs, batch):
    """list of tensors [B,1,1] / [B,1] / [B] (or a packed [B,n(,1)]) -> [B,n]; empty list -> None."""
    if isinstance(inputs, (list, tuple)):
        if len(inputs) == 0:
            return None
        # (the F [B,1,1] views of one packed [B,F,1] block: a slice of it, no F-way concatenation -- two copy launches, 14 us per DeepFM step)
        ts = [t.reshape(batch, -1) for t in merge_packed_views([t for t in inputs])]
        return ts[0] if len(ts) == 1 else torch.cat(ts, dim=1)
    return inputs.reshape(batch, -1)


_CHECK_IDS = os.environ.get("FIL_CHECK_IDS", "0") == "1"   # read once at import


class InnerLayer(Layer):
    """InnerLayer (interactive_layer.py:34-66): pairwise products of the field embeddings.

    call(list of F [B,1,K]) -> list of C(F,2) tensors [B,1,K] in itertools.combinations order, or their sum
    [B,1,K] when use_add.  use_inner=False reads an attribute the reference never defines (``self.dot``, :56,63),
    so it raises AttributeError there; the same error is raised here.
    """

    def __init__(self, use_inner: bool = True, mod=1, seed=2020, perm=None, use_add=False):
        super().__init__()
        self.use_inner = use_inner
        self.mod = mod
        self.seed = seed
        self.perm = perm
        self.use_add = use_add

    def call(self, inputs, **kwargs):
        if not self.use_inner:
            raise AttributeError("'InnerLayer' object has no attribute 'dot' (the reference's use_inner=False branch "
                                 "reads self.dot, which is commented out at interactive_layer.py:56)")
        emb = pack_fields(inputs)
        if self.use_add:
            return Fn.fm(emb, None).unsqueeze(1)
        pairs = Fn.fm_pairs(emb)
        return list(pairs.split(1, dim=1))


class IPnnLayer(Layer):
    """IPnnLayer (interactive_layer.py:68-80): the pair list of InnerLayer(), unchanged."""

    def __init__(self, seed=2020):
        super().__init__()
        self.seed = seed
        self.inner = InnerLayer()

    def call(self, inputs, **kwargs):
        return self.inner(inputs)


class ExtractLayer(Layer):
    """ExtractLayer (interactive_layer.py:82-109): picks, out of a list of per-feature tensors, the ones whose input is named in
    `need_fea`.  `need_inputs` are the model's input descriptors, parallel to the list given to call(); the reference matches
    Keras Input names with their ":0" suffix cut off (`input_.name[:-2]`), so both an object with such a `.name` and a plain
    feature-name string are accepted here.  need_remove=True returns [picked, rest] instead of the picked list.
    compute_mask: None unless mask_zero, then the picked inputs' masks."""

    def __init__(self, need_fea, need_inputs, supports_masking=True, mask_zero=False, need_remove=False):
        super().__init__()
        self.need_fea = need_fea
        self.need_inputs = need_inputs
        self.supports_masking = supports_masking
        self.mask_zero = mask_zero
        self.need_remove = need_remove

    @staticmethod
    def _name(inp):
        if isinstance(inp, str):
            return inp
        n = inp.name
        return n[:-2] if n.endswith(":0") else n

    def call(self, inputs, **kwargs):
        self.need_idx = [i for i, inp in enumerate(self.need_inputs) if self._name(inp) in self.need_fea]
        need_inputs = [inputs[i] for i in self.need_idx]
        if self.need_remove:
            picked = set(self.need_idx)
            return [need_inputs, [t for i, t in enumerate(inputs) if i not in picked]]
        return need_inputs

    def compute_mask(self, inputs, mask=None):
        if not self.mask_zero:
            return None
        return [mask[i] for i in self.need_idx]


class OPnnLayer(Layer):
    """OPnnLayer (interactive_layer.py:111-143).  Its InnerLayer(use_inner=False, ...) reads the attribute the
    reference never defines, so calling it raises AttributeError there; the same here (kept for constructor parity)."""

    def __init__(self, use_reduce=True, seed=2020, use_flatten=True):
        super().__init__()
        self.seed = seed
        self.use_reduce = use_reduce
        self.outer = InnerLayer(use_inner=False, perm=[0, 2, 1], mod=(1, 2))
        self.use_flatten = use_flatten

    def call(self, inputs, **kwargs):
        if self.use_reduce:
            from .core_layer import keras_add
            sum_inputs = keras_add(list(inputs))
            return self.outer([sum_inputs, sum_inputs])
        return self.outer(inputs)


class LinearLayer(Layer):
    """LinearLayer (interactive_layer.py:172-187): w [in,1], b [1] (both `initializer`, default random_normal);
    call(list of tensors [..., in]) -> list of tensordot(t, w, 1) + b."""

    def __init__(self, initializer: str = "random_normal"):
        super().__init__()
        self.initalizer = initializer  # (sic) the reference's attribute name

    def build(self, input_shape):
        last = input_shape[-1]
        while isinstance(last, (list, tuple)):  # Keras passes the list's last shape; input_shape[-1] is its last dim
            last = last[-1]
        self.w = self.add_weight("w", [last, 1], self.initalizer)
        self.b = self.add_weight("b", [1], self.initalizer)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        return [torch.matmul(t, self.w) + self.b for t in inputs]


class AttentionBaseLayer(Layer):
    """AttentionBaseLayer, the AFM pooling (interactive_layer.py:329-366), as coded:
    x = concat(pairs, axis=1) [B,P,K]; score = Dense(1, relu, no bias)(x . single_score_w + single_score_b) [B,P,1];
    ``Activation('softmax')`` normalises over the LAST axis, which has size 1 here, so every weight is exactly 1 and
    the layer returns Dense(output_dim)(sum_p x_p): the reference's quirk is kept (the score weights get zero gradient).
    The pair tensor comes from the HIP pair kernel (InnerLayer); the small dense algebra runs in torch."""

    def __init__(self, attention_dim=4, seed=2020, output_dim=1):
        super().__init__()
        from .core_layer import Dense
        self.atten_dim = attention_dim
        self.seed = seed
        self.output_layer = Dense(output_dim)

    def build(self, input_shape):
        k = input_shape[0][-1]
        self.kernel_w = self.add_weight("single_score_w", [k, self.atten_dim], "glorot_uniform", seed=self.seed)
        self.kernel_b = self.add_weight("single_score_b", [self.atten_dim], "glorot_uniform", seed=self.seed)
        self.single_mlp_kernel = self.add_weight("single_mlp_kernel", [self.atten_dim, 1], "glorot_uniform", seed=self.seed)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        x = pack_fields(inputs)
        score = torch.relu(torch.matmul(torch.matmul(x, self.kernel_w) + self.kernel_b, self.single_mlp_kernel))
        score_w = torch.softmax(score, dim=-1)
        atten_inputs = (score_w * x).sum(dim=1)
        return self.output_layer(atten_inputs)


class FmLayer(Layer):
    """FmLayer (interactive_layer.py:145-170): call([cross_embed, linear_embed]) -> [B,1,K] =
    sum_{i<j} e_i*e_j + the broadcast sum of the linear terms (no reduction over K)."""

    def __init__(self, use_inner: bool = True, mod=1, use_add=True, **kwargs):
        super().__init__(**kwargs)
        self.cross = InnerLayer(use_inner=use_inner, mod=mod, use_add=use_add)
        self.use_add = use_add

    def call(self, inputs, **kwargs):
        cross_embed, linear_embed = inputs
        if not self.use_add:
            # reference: Add([list_of_pairs] + linear) -> Keras raises on the nested list
            raise ValueError("FmLayer(use_add=False): the reference passes a nested list to keras Add and fails")
        if not self.cross.use_inner:
            return self.cross(cross_embed)  # raises like the reference
        emb = pack_fields(cross_embed)
        lin = pack_linear(linear_embed, emb.shape[0])
        return Fn.fm(emb, lin).unsqueeze(1)


class CrossLayer(Layer):
    """DCN cross network (interactive_layer.py:250-282).  build creates outer_weight_i / outer_bias_i [D,1]
    (glorot_uniform(seed) / zeros); call(x [B,D]) -> [B,D,1] (not squeezed, like the reference)."""

    def __init__(self, cross_hidden=3, seed=2020, **kwargs):
        super().__init__(**kwargs)
        self.cross_hidden = cross_hidden
        self.seed = seed

    def build(self, input_shape):
        d = input_shape[-1]
        self.kernel = [self.add_weight("outer_weight_{}".format(i), [d, 1], "glorot_uniform", seed=self.seed)
                       for i in range(self.cross_hidden)]
        self.bias = [self.add_weight("outer_bias_{}".format(i), [d, 1], "zeros") for i in range(self.cross_hidden)]
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        w = torch.cat([k.t() for k in self.kernel], dim=0)  # [L,D]
        b = torch.cat([k.t() for k in self.bias], dim=0)
        if not self.fits_kernel_menu(inputs):
            return self._composed(inputs, w, b).unsqueeze(-1)
        return Fn.dcn_cross(inputs, w, b).unsqueeze(-1)

    def fits_kernel_menu(self, x):
        """fil_dcn_*'s limit (include/fil.h, csrc/dcn.hip): cross_hidden <= 16, any D (D <= 4096 with cross_hidden <= 6 and the
        parameters within the LDS run the register-resident kernels, anything else the generic two-pass ones).  The reference has no
        limit (:255-282): a deeper layer takes the composed path below instead of raising."""
        return self.cross_hidden <= 16

    def _composed(self, x, w, b):
        """The reference's recurrence (:275-282) on the GPU with plain torch ops (autograd for the backward), layer by layer:
        s_l = K.dot(x_l^T, w_l)  [B,1];  x_{l+1} = K.batch_dot(x0, s_l) + x_l + b_l.  Five passes over [B,D] per layer where the fused
        kernel makes two for the whole stack -- it exists so that a reference-legal layer never raises."""
        Fn._require_cuda(x)          # (torch ops would run on a CPU tensor: there is no CPU path in this package)
        x0 = x.to(torch.float32)
        xl = x0
        for l in range(self.cross_hidden):
            s = torch.matmul(xl, w[l].unsqueeze(-1))                # [B,1]
            xl = x0 * s + xl + b[l]
        return xl


class CIN(Layer):
    """xDeepFM compressed interaction network (interactive_layer.py:285-327).

    build creates one Conv1D(size, 1) per entry of conv_size (kernel [1, H_{l-1}*F, H_l] glorot_uniform, bias [H_l]
    zeros; channel c = h*F+f) and, when output_dim == 1, the Dense(1) head (kernel [L*K,1], bias [1]).
    call(x [B,F,K]) -> [B,1] (or the pooled concat [B, L*K] when output_dim != 1).  The sum-pool is over the
    feature-map axis (reference :322)."""

    def __init__(self, conv_size=None, output_dim=1, mode=0):
        super().__init__()
        if conv_size is None:
            conv_size = [200, 200, 200]
        self.conv_size = list(conv_size)
        self.output_dim = output_dim
        self.mode = mode

    def build(self, input_shape):
        _, f, k = input_shape
        self.conv_kernels, self.conv_biases = [], []
        hp = f
        for l, h in enumerate(self.conv_size):
            self.conv_kernels.append(self.add_weight("hidden_conv_{}_kernel".format(l), [1, hp * f, h], "glorot_uniform"))
            self.conv_biases.append(self.add_weight("hidden_conv_{}_bias".format(l), [h], "zeros"))
            hp = h
        if self.output_dim == 1:
            self.logit_kernel = self.add_weight("logit_layer_kernel", [len(self.conv_size) * k, 1], "glorot_uniform")
            self.logit_bias = self.add_weight("logit_layer_bias", [1], "zeros")
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        x = pack_fields(inputs)
        # SparseEmbed(emit_xt=True) leaves the block's [B*K, F] transpose on it: the kernels then read that in place
        xt = getattr(x, "_fil_xt", None)
        if xt is not None and (x.dim() != 3 or tuple(xt.shape) != (x.shape[0] * x.shape[2], x.shape[1]) or xt.device != x.device
                               or getattr(x, "_fil_xt_version", None) != x._version):
            xt = None       # (version: the block was edited in place after the gather -- mask multiply, dropout_ -- xt is stale)
        # ([1, C, H] Conv1D kernels as [C, H]: a VIEW -- `w[0]` is a select, whose backward zero-fills a [1, C, H] buffer and copies the
        # gradient into it: six launches and 28 us per xDeepFM step)
        Ws = [w.view(w.shape[1], w.shape[2]) for w in self.conv_kernels]
        if not self.fits_kernel_menu(x):
            return self._composed(x, Ws)
        if self.output_dim == 1:
            return Fn.cin(x, Ws, self.conv_biases, self.logit_kernel, self.logit_bias, output_dim=1, mode=self.mode, xt=xt)
        return Fn.cin(x, Ws, self.conv_biases, None, None, output_dim=self.output_dim, mode=self.mode, xt=xt)

    def fits_kernel_menu(self, x):
        """fil_cin_*'s limits (include/fil.h): F <= 64, H_l <= 256, L <= 8.  The reference has none (:296-327): a layer outside them
        takes the composed path below instead of raising."""
        return x.shape[1] <= 64 and max(self.conv_size) <= 256 and len(self.conv_size) <= 8

    def _composed(self, x, Ws, rows_per_chunk=1 << 16):
        """The reference's op graph (:310-327) on the GPU for shapes outside the HIP kernels' menu: per layer the outer product
        z[b,k,(h,f)] = x^{l-1}[b,h,k] x[b,f,k] and one library GEMM with the Conv1D kernel (plain rocBLAS through torch, autograd for
        the backward), in chunks of samples so that the materialised z stays bounded.  Slow next to the kernels -- it exists so
        that a reference-legal layer never raises."""
        Fn._require_cuda(x)          # (torch ops would run on a CPU tensor: there is no CPU path in this package)
        B, F, K = x.shape
        cmax = max([F] + self.conv_size[:-1]) * F
        step = max(1, rows_per_chunk // max(1, K * cmax // 64))
        outs = []
        for lo in range(0, B, step):
            xc = x[lo:lo + step]
            xk = xc.permute(0, 2, 1)                                # [b,K,F]
            pre, pools = xk, []
            for w, bias in zip(Ws, self.conv_biases):
                z = (pre.unsqueeze(3) * xk.unsqueeze(2)).reshape(xc.shape[0], K, -1)     # channel c = h*F + f (:316-318)
                pre = torch.matmul(z, w) + bias                     # Conv1D(size, 1) (:319)
                pools.append(pre.sum(-1))                           # reduce_sum over the feature maps (:322)
            outs.append(torch.cat(pools, 1))
        pooled = torch.cat(outs, 0)
        if self.output_dim == 1:
            return torch.matmul(pooled, self.logit_kernel) + self.logit_bias
        return pooled


class SparseEmbed(Layer):
    """SparseEmbed (interactive_layer.py:189-247): one embedding table per sparse field.

    sparse_info: list of descriptors with .fea_name, .word_size and .cross_unit (embedding dim; .linear_unit when
    is_linear), as the reference's sparseFea namedtuple (data_prepare.py:59).  All tables live in one concatenated
    parameter so that one kernel launch gathers the packed [B,F,K] block (bit-exact row copies).
    call(list of F integer tensors [B,1] or a packed [B,F]) -> the reference's list of F tensors [B,1,K]
    (views of the packed block; ``packed=True`` returns the block itself), flattened / summed like the reference
    when use_flatten / use_add.

    Descriptor fields honoured like the reference's Embedding arguments (:209-218), cross tables only:
      pre_weight    initial table values ([V_f, K] array, or Keras' one-element list of it)
      is_trainable  False freezes the field: its rows get no gradient
      emb_reg       l2(emb_reg) on the field's whole table -> regularization_losses() (added to the loss by the trainer)
    Ids are range-checked in the kernel: an id outside [0, word_size) produces a zero row and no gradient (Keras on a GPU);
    ``check_ids=True`` (or FIL_CHECK_IDS=1) additionally raises, like Keras on the CPU does (costs a device sync per call).
    ``sparse_grad=True`` hands the table gradient out as a sparse COO tensor over the touched rows (Keras' IndexedSlices)
    instead of a dense table; either way it is deterministic (sorted segment sums, no atomics)."""

    def __init__(self, sparse_info: list, is_linear=False, use_flatten=True, use_add=False, seed=2020, support_masking=True,
                 mask_zero=False, packed=False, check_ids=None, sparse_grad=False, emit_xt=False, out_dtype=None):
        super().__init__()
        self.sparse_info = sparse_info
        self.is_linear = is_linear
        self.use_flatten = use_flatten
        self.use_add = use_add
        self.seed = seed
        self.supports_masking = support_masking
        self.mask_zero = mask_zero
        self.packed = packed
        self.check_ids = _CHECK_IDS if check_ids is None else bool(check_ids)
        self.sparse_grad = sparse_grad
        self.emit_xt = emit_xt      # extension: also emit the block transposed to [B*K, F] for a CIN consumer (fil_embed_gather_xt)
        self.out_dtype = out_dtype  # extension: torch.bfloat16 = the block in bf16 straight out of the gather (a bf16 model's cast, fused)

    def build(self, input_shape):
        dims = {int(i.linear_unit if self.is_linear else i.cross_unit) for i in self.sparse_info}
        if len(dims) != 1 or 0 in dims:
            raise ValueError("SparseEmbed on the HIP path needs one common non-zero embedding dim, got %s" % sorted(dims))
        k = dims.pop()
        sizes = [int(i.word_size) for i in self.sparse_info]
        self.embeddings = self.add_weight("embeddings", [sum(sizes), k], "zeros")
        off = 0
        offsets, frozen, self._reg = [], [], []
        for n, (v, info) in enumerate(zip(sizes, self.sparse_info)):
            # each table is initialised like its own Keras Embedding (glorot_uniform(seed) over [V_f, K];
            # linear tables use Keras' default 'uniform' = U(-0.05, 0.05))
            pre = None if self.is_linear else getattr(info, "pre_weight", None)
            if pre is not None:
                w = pre[0] if isinstance(pre, (list, tuple)) else pre       # Keras: weights=[table]
                w = torch.as_tensor(w, dtype=torch.float32)
                if tuple(w.shape) != (v, k):
                    raise ValueError("pre_weight of %s must be [%d,%d], got %s" % (info.fea_name, v, k, tuple(w.shape)))
                with torch.no_grad():
                    self.embeddings[off:off + v].copy_(w)
            elif self.is_linear:
                with torch.no_grad():
                    self.embeddings[off:off + v].uniform_(-0.05, 0.05)
            else:
                glorot_uniform_(self.embeddings.data[off:off + v], seed=self.seed)
            trainable = True if self.is_linear else getattr(info, "is_trainable", True)
            frozen.append(0 if (trainable is None or trainable) else 1)
            reg = 0.0 if self.is_linear else (getattr(info, "emb_reg", 0.0) or 0.0)
            if reg:
                self._reg.append((off, off + v, float(reg)))
            offsets.append(off)
            off += v
        dev = self._build_device
        self.register_buffer("offsets", torch.tensor(offsets, dtype=torch.int64, device=dev))
        self.register_buffer("sizes", torch.tensor(sizes, dtype=torch.int64, device=dev))
        self.register_buffer("frozen", torch.tensor(frozen, dtype=torch.uint8, device=dev) if any(frozen) else None)
        # tables with the same field layout (the embeddings and the linear weights of one model) share the sort of their gradient
        self._layout_key = (tuple(sizes), tuple(frozen) if any(frozen) else None)
        super().build(input_shape)

    def regularization_losses(self):
        """tf.keras.regularizers.l2(emb_reg) on each field's table (interactive_layer.py:217): emb_reg * sum(table^2)."""
        if not self.built:
            return []
        return [reg * self.embeddings[lo:hi].square().sum() for lo, hi, reg in self._reg]

    def table_l2_ranges(self):
        """[(row_lo, row_hi, emb_reg)] of the regularised fields of the concatenated table (for dp.add_table_l2_grad_)."""
        return list(self._reg) if self.built else []

    def call(self, inputs, **kwargs):
        idx = torch.cat([t.reshape(t.shape[0], 1) for t in inputs], dim=1) if isinstance(inputs, (list, tuple)) else inputs
        idx = idx.to(torch.int64)
        oob = torch.zeros((), dtype=torch.int32, device=idx.device) if self.check_ids else None
        block = Fn.embed_gather(self.embeddings, self.offsets, idx, sizes=self.sizes, frozen=self.frozen,
                                sparse_grad=self.sparse_grad, oob_count=oob, layout_key=self._layout_key,
                                emit_xt=self.emit_xt, out_dtype=self.out_dtype)  # [B,F,K]
        if oob is not None and int(oob) > 0:
            bad = ((idx < 0) | (idx >= self.sizes)).nonzero()[0].tolist()
            raise IndexError("SparseEmbed: %d ids outside their vocabulary, first at sample %d, field %s (id %d, word_size %d)"
                             % (int(oob), bad[0], self.sparse_info[bad[1]].fea_name, int(idx[bad[0], bad[1]]),
                                int(self.sizes[bad[1]])))
        if self.packed:
            return block
        embed_list = list(block.split(1, dim=1))  # F x [B,1,K]
        if self.use_flatten:
            embed_list = [e.reshape(e.shape[0], -1) for e in embed_list]
        if self.use_add:
            # Keras Add over the F per-field tensors (tf.add_n: order unspecified) as ONE reduction of the packed block -- the
            # Python loop of F-1 additions was 2(F-1) tiny kernels per step (xDeepFM's linear terms: 165 us of a 2.3 ms step)
            embed_list = block.sum(dim=1) if self.use_flatten else block.sum(dim=1, keepdim=True)
        if self.mask_zero:
            masks = [(idx[:, f:f + 1] != 0) for f in range(idx.shape[1])]
            return embed_list, masks
        return embed_list
