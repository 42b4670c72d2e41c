"""Glue layers around the interaction kernels: the re-host of the reference's core_layer.py
(StackLayer :32, ScoreLayer :58, MergeScoreLayer :86, HiddenLayer :102, ResActivateLayer :131, DnnLayer :159).
Dense layers are plain GEMMs and go through torch (hipBLASLt); the AutoInt wrapper path of DnnLayer dispatches to
the fused attention kernel."""
import warnings

import torch

from .. import functional as F
from .base import Layer, merge_packed_views, glorot_uniform_
from .behavior_layer import MultHeadAttentionLayer
from .interactive_layer import InnerLayer


def keras_add(tensors):
    """tf.keras.layers.Add: left-to-right sum; raises ValueError when shapes are not broadcast-compatible or an
    element is not a tensor (DnnLayer relies on that to skip the residual, core_layer.py:211-214)."""
    out = None
    for t in tensors:
        if not torch.is_tensor(t):
            raise ValueError("Add expects tensors")
        if out is None:
            out = t
            continue
        try:
            torch.broadcast_shapes(out.shape, t.shape)
        except RuntimeError as e:
            raise ValueError(str(e))
        out = out + t
    if out is None:
        raise ValueError("Add of an empty list")
    return out


class Dense(Layer):
    """tf.keras.layers.Dense(units, activation=None): kernel [in, units] glorot_uniform, bias [units]."""

    def __init__(self, units, activation=None, seed=None, bias_initializer="zeros"):
        super().__init__()
        self.units = units
        self.activation = activation
        self.seed = seed
        self.bias_initializer = bias_initializer

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", [input_shape[-1], self.units], "glorot_uniform", seed=self.seed)
        self.bias = self.add_weight("bias", [self.units], self.bias_initializer, seed=self.seed)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        if (inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32 and self.kernel.dtype == torch.float32
                and not torch.is_autocast_enabled("cuda") and inputs.shape[0] > 0):
            y = F.dense(inputs, self.kernel, self.bias)     # fp32 on the GPU: the library's own GEMM both ways (functional._DenseFn)
        else:
            y = torch.matmul(inputs, self.kernel) + self.bias
        if self.activation == "softmax":
            y = torch.softmax(y, dim=-1)
        elif self.activation == "sigmoid":
            y = torch.sigmoid(y)
        elif self.activation is not None:
            y = self.activation(y)
        return y


class StackLayer(Layer):
    """Flatten each input (use_flat) and concatenate along `axis` (default last); a single input passes through."""

    def __init__(self, use_flat=True, axis=None):
        super().__init__()
        self.use_flat = use_flat
        self.axis = axis if axis else -1  # the reference treats axis=None/0 as the Concatenate default (-1)

    def call(self, inputs, **kwargs):
        inputs = list(inputs)
        if self.use_flat or self.axis == 1:
            # F split views of one packed [B,F,K] block concatenate back into a slice of it: no copy
            inputs = merge_packed_views(inputs)
        if self.use_flat:
            inputs = [t.reshape(t.shape[0], -1) for t in inputs]
        if len(inputs) == 1:
            return inputs[0]
        if inputs[0].is_cuda and torch.is_autocast_enabled("cuda"):
            # under autocast `cat` promotes mixed inputs to the widest type (fp32) and the layer behind it casts the result down
            # again: cast the wide pieces first instead -- the same values, one narrow concatenation instead of a wide one + a cast
            lo = torch.get_autocast_dtype("cuda")
            if any(t.dtype == lo for t in inputs) and all(t.dtype in (lo, torch.float32) for t in inputs):
                inputs = [t.to(lo) for t in inputs]
        return torch.cat(inputs, dim=self.axis)


class ScoreLayer(Layer):
    def __init__(self, use_add=False, use_inner=False, use_global=False, seed=2020):
        super().__init__()
        self.use_add = use_add
        self.use_inner = use_inner
        self.inner = InnerLayer(use_inner=True)
        self.use_global = use_global
        self.seed = seed

    def build(self, input_shape):
        if self.use_global:
            self.global_bias = self.add_weight("global_bias", (1,), "glorot_uniform", seed=self.seed)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        if self.use_add and not self.use_global and not self.use_inner:
            parts = list(inputs)
            # the common head (xDeepFM, NFM-style sums of [B,1] scores): Add + sigmoid as ONE launch each way
            if (1 <= len(parts) <= 4 and all(t.is_cuda and t.dtype == torch.float32 and t.shape == parts[0].shape for t in parts)):
                return F.score_add_sigmoid(parts)
        if self.use_add:
            inputs = keras_add(list(inputs))
            if self.use_global:
                inputs = keras_add([inputs, self.global_bias])
        if self.use_inner:
            inputs = self.inner(inputs)
        return torch.sigmoid(inputs)


class MergeScoreLayer(Layer):
    def __init__(self, use_merge: bool = True, output_dim=2):
        super().__init__()
        self.concat = StackLayer()
        self.dense = Dense(units=output_dim, activation="softmax")
        self.use_merge = use_merge

    def call(self, inputs, **kwargs):
        fused = self._fused(inputs)
        if fused is not None:
            return fused
        if self.use_merge:
            inputs = self.concat(inputs)
        return self.dense(inputs)

    def _fused(self, inputs):
        """StackLayer concat -> Dense(softmax) as ONE launch each way (functional.merge_softmax, csrc/head.hip) when the inputs are what
        the zoo hands over: 1..4 CUDA tensors, each fp32 or bf16 (autocast), that flatten to [B, w].  Anything
        else -- and the first call, which builds the Dense -- takes the composed path."""
        if not self.dense.built or self.dense.units > 8:
            return None
        parts = list(inputs) if self.use_merge else [inputs]
        if not 1 <= len(parts) <= 4 or not all(torch.is_tensor(t) and t.is_cuda and t.dim() >= 2 for t in parts):
            return None
        if self.use_merge and self.concat.axis not in (-1, 1):
            return None
        parts = [t.reshape(t.shape[0], -1) for t in parts]
        if any(t.dtype not in (torch.float32, torch.bfloat16) or t.shape[0] != parts[0].shape[0] for t in parts):
            return None
        if sum(t.shape[1] for t in parts) != self.dense.kernel.shape[0] or self.dense.kernel.shape[0] > 8192:
            return None
        return F.merge_softmax(parts, self.dense.kernel, self.dense.bias)


class HiddenLayer(Layer):
    """Dense(hidden_units) with glorot_uniform(seed) kernel AND bias and kernel_regularizer=l2(l2_reg) (core_layer.py:111-116),
    optional BatchNorm; call returns (x, inputs).  other_dense replaces the Dense (e.g. by a MultHeadAttentionLayer).
    The BatchNorm module is created in build() (never inside call): an optimizer or .to(device) made after the first
    build sees every parameter."""

    def __init__(self, hidden_units: int, use_bn: bool = True, seed=2020, l2_reg=0, other_dense=None):
        super().__init__()
        self.dense = other_dense if other_dense else Dense(hidden_units, seed=seed, bias_initializer="glorot_uniform")
        self.hidden_units = hidden_units
        self.use_bn = use_bn
        self.l2_reg = l2_reg
        self.bn = None

    def build(self, input_shape):
        if self.use_bn:
            width = self.hidden_units if isinstance(self.dense, Dense) else getattr(self.dense, "attention_dim", None)
            if width is None:
                raise ValueError("HiddenLayer(use_bn=True) around %s: cannot tell the output width at build time" % type(self.dense).__name__)
            self.bn = torch.nn.BatchNorm1d(int(width), eps=1e-3, momentum=0.01).to(self._build_device)
        super().build(input_shape)

    def regularization_losses(self):
        """tf.keras.regularizers.l2(l2_reg) on the Dense kernel (core_layer.py:113)."""
        if not self.l2_reg or not isinstance(self.dense, Dense) or not self.dense.built:
            return []
        return [float(self.l2_reg) * self.dense.kernel.square().sum()]

    def call(self, inputs, **kwargs):
        x = self.dense(inputs)
        if self.use_bn:
            x = self.bn(x)
        return x, inputs


class ResActivateLayer(Layer):
    """BatchNorm / LayerNorm (modules created in build, from the input width) followed by the activation
    (core_layer.py:131-156)."""

    def __init__(self, use_bn, use_ln, hidden_activate):
        super().__init__()
        self.use_bn = use_bn
        self.use_ln = use_ln
        self.active = hidden_activate
        self.bn = None
        self.ln = None

    def build(self, input_shape):
        width = int(input_shape[-1])
        if self.use_bn:
            self.bn = torch.nn.BatchNorm1d(width, eps=1e-3, momentum=0.01).to(self._build_device)
        if self.use_ln:
            self.ln = torch.nn.LayerNorm(width, eps=1e-3).to(self._build_device)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        if self.use_bn:
            inputs = self.bn(inputs)
        if self.use_ln:
            inputs = self.ln(inputs)
        return self.active(inputs)


class DnnLayer(Layer):
    """DnnLayer (core_layer.py:159-226): a stack of hidden layers with a residual Add every `res_unit` layers
    (skipped when the shapes do not broadcast, :211-214), an activation block per layer and an optional logit head.

    AutoInt builds DnnLayer(res_unit=1, other_dense=[MultHeadAttentionLayer]): that path is one fused HIP kernel
    (projection, sigmoid attention, LayerNorm, residual add, ReLU)."""

    def __init__(self, hidden_units: list = None, l2_reg=0, hidden_activate=None, use_bn: bool = False, res_unit=1,
                 output_dim=-1, seed=2020, other_dense=None, use_ln: bool = False, use_flatten=False, **kwargs):
        super().__init__(**kwargs)
        if hidden_activate is None:
            hidden_activate = torch.nn.ReLU()
        self.hidden_list = other_dense
        if not other_dense:
            # The reference builds HiddenLayer(hidden_units=dim, use_bn=False, other_dense=other_dense) (core_layer.py:182):
            # its own l2_reg argument never reaches the hidden layers, so nothing is regularised.  Same here, said aloud.
            if l2_reg:
                warnings.warn("DnnLayer(l2_reg=%r): the reference drops this argument (core_layer.py:182 builds its HiddenLayers "
                              "without it); no regulariser is applied. Use HiddenLayer(l2_reg=...) to get one." % (l2_reg,))
            self.hidden_list = [HiddenLayer(hidden_units=dim, use_bn=False, other_dense=other_dense) for dim in hidden_units]
        self.hidden_modules = torch.nn.ModuleList(self.hidden_list)
        self.hidden_activate = hidden_activate
        self.activate = torch.nn.ModuleList(
            [ResActivateLayer(use_bn=use_bn, use_ln=use_ln, hidden_activate=hidden_activate) for _ in self.hidden_list])
        self.seed = 2020
        self.output_dim = output_dim
        self.res_unit = res_unit
        self.use_bn = use_bn
        self.use_ln = use_ln
        if output_dim != -1:
            self.logit_layer = Dense(units=output_dim, seed=seed, bias_initializer="glorot_uniform")
        self.use_flatten = use_flatten

    def _fusable(self):
        return (len(self.hidden_list) == 1 and isinstance(self.hidden_list[0], MultHeadAttentionLayer)
                and self.res_unit == 1 and not self.use_bn and not self.use_ln
                and isinstance(self.hidden_activate, torch.nn.ReLU) and not self.hidden_list[0].head_concat
                and self.hidden_list[0].attention_head_dim > 1)

    def _dense_relu(self, idx_, hidden_layer, x):
        """A plain hidden layer whose residual Add is skipped (res_unit == 1 and in != units: keras Add raises on the shapes, :211-214)
        is relu(x @ W + b): one library GEMM with the bias + ReLU in its epilogue, one HIP pass + two GEMMs backward
        (functional.dense_relu).  None: anything else (first call: the Dense is not built yet; norms; a square layer, whose residual
        IS added; another activation; a CPU tensor) takes the composed path below."""
        d = hidden_layer.dense if isinstance(hidden_layer, HiddenLayer) else None
        if (not isinstance(d, Dense) or not d.built or d.activation is not None or hidden_layer.use_bn or self.use_bn or self.use_ln
                or self.res_unit != 1 or type(self.hidden_activate) is not torch.nn.ReLU or not torch.is_tensor(x) or not x.is_cuda
                or x.dim() != 2 or x.shape[1] != d.kernel.shape[0] or d.kernel.shape[0] == d.kernel.shape[1]
                or x.dtype not in (torch.float32, torch.bfloat16)):
            return None
        return F.dense_relu(x, d.kernel, d.bias)

    def call(self, inputs, **kwargs):
        x = inputs
        if self._fusable():
            layer = self.hidden_list[0]
            if not layer.built:
                layer._build_device = x.device
                layer.build(tuple(x.shape))
                layer.built = True
            x = layer.fused_relu(x)
        else:
            res = [[], []]
            for idx_, hidden_layer in enumerate(self.hidden_list):
                fused = self._dense_relu(idx_, hidden_layer, x)
                if fused is not None:
                    x = fused
                    res = [x, x]
                    continue
                x, ori = hidden_layer(x)
                if idx_ == 0:
                    res = [ori, x]
                if (idx_ + 1) % self.res_unit != 0 or self.res_unit == 1:
                    res[-1] = x
                if (idx_ + 1) % self.res_unit == 0:
                    try:
                        x = keras_add(res)
                    except ValueError:
                        x = res[-1]
                x = self.activate[idx_](x)
                if (idx_ + 1) % self.res_unit == 0:
                    res[0] = x
        if self.use_flatten:
            x = x.reshape(x.shape[0], -1)
        if self.output_dim != -1:
            x = self.logit_layer(x)
        return x


class IntraViewPoolingLayer(Layer):
    """IntraViewPoolingLayer (core_layer.py:228-238): mean over axis 1, kept as a size-1 axis -- [B,N,...] -> [B,1,...]."""

    def __init__(self):
        super().__init__()

    def call(self, inputs, **kwargs):
        return torch.mean(inputs, dim=1).unsqueeze(1)


class AlignLayer(Layer):
    """AlignLayer (core_layer.py:240-258): a list of tensors whose last dims differ is brought to the LARGEST last dim -- every
    narrower input goes through its own Dense(max_dim) (no activation, glorot kernel, zero bias), the widest ones pass through
    untouched.  build() creates `format_dense` = [Dense or None per input], like the reference."""

    def __init__(self):
        super().__init__()

    def build(self, input_shape):
        dim_list = [int(sh[-1]) for sh in input_shape]
        max_dim = max(dim_list)
        self.format_dense = [Dense(units=max_dim) if d < max_dim else None for d in dim_list]
        for i, fd in enumerate(self.format_dense):
            if fd is not None:
                self.add_module("format_dense_%d" % i, fd)
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        return [fd(x) if fd is not None else x for x, fd in zip(inputs, self.format_dense)]
