"""Keras-style layer protocol on torch.nn.Module.

The reference's only "plugin API" is the tf.keras.layers.Layer protocol: ``__init__`` stores
hyper-parameters, ``build(input_shape)`` runs lazily on the first call and creates weights with
``add_weight(name, shape, initializer)``, ``call(inputs, **kwargs)`` is pure.  This base class reproduces that
lifecycle so the re-hosted layers keep the reference's constructor kwargs, weight names and call signatures.
"""
import math

import torch


def shape_of(inputs):
    """Keras-like input_shape: a tuple for a tensor, a list of shapes for a list of tensors (nested lists allowed)."""
    if isinstance(inputs, (list, tuple)):
        return [shape_of(t) for t in inputs]
    if inputs is None:
        return None
    return tuple(inputs.shape)


def first_tensor(inputs):
    if isinstance(inputs, (list, tuple)):
        for t in inputs:
            r = first_tensor(t)
            if r is not None:
                return r
        return None
    return inputs


def merge_packed_views(tensors):
    """The reference passes F separate [B,1,K] tensors around; here they are usually the `split(1, dim=1)` views of ONE
    packed [B,F,K] block (SparseEmbed).  Runs of such adjacent views are replaced by a single slice of their base, so that
    re-packing them is a view instead of an F-way concatenation copy (same values, same autograd graph endpoints)."""
    out, i, n = [], 0, len(tensors)
    while i < n:
        t = tensors[i]
        base = getattr(t, "_base", None)
        # the same for the [B,1] column views `dense[:, i:i+1]` of one contiguous [B,n] block (FeatureInput's dense inputs)
        if (base is not None and base.dim() == 2 and t.dim() == 2 and t.shape[1] == 1 and t.shape[0] == base.shape[0]
                and base.is_contiguous() and t.stride() == base.stride()):
            first = t.storage_offset() - base.storage_offset()
            j = i + 1
            while (j < n and getattr(tensors[j], "_base", None) is base and tensors[j].shape == t.shape and tensors[j].stride() == t.stride()
                   and tensors[j].storage_offset() == base.storage_offset() + first + (j - i)):
                j += 1
            if j - i > 1 and 0 <= first and first + (j - i) <= base.shape[1]:
                out.append(base if (first == 0 and j - i == base.shape[1]) else base.narrow(1, first, j - i))
                i = j
                continue
        ok = (base is not None and base.dim() == 3 and t.dim() == 3 and t.shape[1] == 1 and t.shape[0] == base.shape[0]
              and t.shape[2] == base.shape[2] and base.is_contiguous() and t.stride() == base.stride())
        if not ok:
            out.append(t)
            i += 1
            continue
        k = base.shape[2]
        first = (t.storage_offset() - base.storage_offset()) // k
        j = i + 1
        while (j < n and getattr(tensors[j], "_base", None) is base and tensors[j].shape == t.shape
               and tensors[j].stride() == t.stride()
               and tensors[j].storage_offset() == base.storage_offset() + (first + (j - i)) * k):
            j += 1
        if j - i == 1 or (t.storage_offset() - base.storage_offset()) % k != 0:
            out.append(t)
            i += 1
        else:
            out.append(base if (first == 0 and j - i == base.shape[1]) else base.narrow(1, first, j - i))
            i = j
    return out


def glorot_uniform_(tensor, seed=None):
    """Keras glorot_uniform: U(-l, l), l = sqrt(6 / (fan_in + fan_out)); rank>2 uses the receptive-field rule.
    (TF's RNG stream cannot be matched; parity tests pass explicit weights.)"""
    shape = tuple(tensor.shape)
    if len(shape) < 1:
        fan_in = fan_out = 1
    elif len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = 1
        for d in shape[:-2]:
            rf *= d
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    gen = None
    if seed is not None:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(int(seed))
    with torch.no_grad():
        vals = (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * lim
        tensor.copy_(vals.to(tensor.device))
    return tensor


class Layer(torch.nn.Module):
    """tf.keras.layers.Layer lifecycle: lazy build() on first __call__, weights via add_weight()."""

    def __init__(self, name=None, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError("unexpected keyword arguments: %s" % sorted(kwargs))
        self.built = False
        self.layer_name = name
        self._build_device = None

    # -- Keras protocol -------------------------------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def add_weight(self, name, shape, initializer="glorot_uniform", trainable=True, seed=None):
        shape = tuple(int(s) for s in shape)
        p = torch.nn.Parameter(torch.empty(shape, dtype=torch.float32, device=self._build_device), requires_grad=trainable)
        if callable(initializer):
            initializer(p)
        elif initializer in ("zeros", "zero"):
            torch.nn.init.zeros_(p)
        elif initializer in ("ones", "one"):
            torch.nn.init.ones_(p)
        elif initializer == "glorot_uniform":
            glorot_uniform_(p, seed=seed)
        elif initializer == "random_normal":
            torch.nn.init.normal_(p, mean=0.0, std=0.05)  # Keras RandomNormal default stddev
        else:
            raise ValueError("unknown initializer %r" % (initializer,))
        self.register_parameter(name, p)
        return p

    # -- torch glue -----------------------------------------------------------------------------------------
    def forward(self, inputs, **kwargs):
        if not self.built:
            t = first_tensor(inputs)
            self._build_device = t.device if t is not None else None
            self.build(shape_of(inputs))
            self.built = True
        return self.call(inputs, **kwargs)

    def compute_mask(self, inputs, mask=None):
        return mask

    # -- regularisation losses (Keras: kernel_regularizer / embeddings_regularizer contribute to model.losses) ----------
    def regularization_losses(self):
        """Scalar tensors this layer adds to the training loss (overridden where the reference attaches a regularizer)."""
        return []


def collect_regularization_loss(module, skip_tables=False):
    """Sum of the regularisation terms of every Layer below `module` (what Keras adds to the compiled loss as
    sum(model.losses)); a zero scalar when nothing is regularised.  skip_tables=True leaves out the l2 terms of embedding
    tables (layers with `table_l2_ranges`): a data-parallel trainer adds their gradient analytically AFTER the sparse row
    exchange (dp.add_table_l2_grad_), because an autograd term over a whole table makes its gradient dense and breaks the
    exchange's "zero outside the touched rows" contract."""
    terms = []
    for m in module.modules():
        if isinstance(m, Layer):
            if skip_tables and hasattr(m, "table_l2_ranges"):
                continue
            terms += list(m.regularization_losses())
    if not terms:
        p = next(module.parameters(), None)
        return torch.zeros((), device=p.device if p is not None else None)
    return torch.stack([t.reshape(()) for t in terms]).sum()
