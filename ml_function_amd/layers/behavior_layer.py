"""AutoInt's interacting layer: the MI355X re-host of ProductAttentionLayer (behavior_layer.py:272-311) and
MultHeadAttentionLayer (:313-380) of the reference.  Reference quirks are kept on purpose: the "softmax" is a
sigmoid (:286,308), V is projected with key_w (:360; value_w exists but never receives a gradient), LayerNorm
epsilon is Keras' default 1e-3, the output is head-major [H,B,F,A]."""
import torch

from .. import functional as Fn
from .._lib import FilError
from .base import Layer


class ProductAttentionLayer(Layer):
    """Scaled product attention with a sigmoid in place of the softmax (behavior_layer.py:272-311).

    call([q, k, v], mask=None) -> sigmoid(q k^T [/ sqrt(A)] [masked]) v on the HIP path (fil_pattn_*: the scores never
    leave registers).  mask_mod 1 right-multiplies the scores by the float mask (:300-302), mask_mod 2 adds
    mask * (-100000) (:303-306).  Inside MultHeadAttentionLayer the same arithmetic runs fused with the projections,
    LayerNorm and residual (fil_attn_*) when there is no mask."""

    def __init__(self, use_scale=False, supports_masking=True, mask_mod=1):
        super().__init__()
        self.use_scale = use_scale
        self.supports_masking = supports_masking
        self.mask_mod = mask_mod

    def call(self, inputs, mask=None, **kwargs):
        q, k, v = inputs
        return Fn.product_attention(q, k, v, use_scale=self.use_scale, mask=mask, mask_mod=self.mask_mod)


class MultHeadAttentionLayer(Layer):
    """MultHeadAttentionLayer (behavior_layer.py:313-380).

    build creates query_w, key_w, value_w, res_w, each [K_in, head, dim] (glorot_uniform(seed); res_w uses Keras'
    default initializer, also glorot_uniform) and the LayerNormalization gamma/beta [dim].
    call(x [B,F,K], mask=None) -> [atten_v, res], both [H,B,F,A] (atten_v is [B,H,F,A] when head_concat; a single
    squeezed tensor when attention_head_dim == 1, like the reference :374-375)."""

    def __init__(self, attention_dim, attention_head_dim, seed=2020, use_scale=True, use_res=True, use_ln=True,
                 head_concat=False, supports_masking=True, atten_mask_mod=1, precision="f32"):
        super().__init__()
        self.precision = precision  # extension: "f16_mfma" = BASELINE config 5 (fp16 MFMA products, fp32 accumulate)
        self.attention_dim = attention_dim
        self.attention_head_dim = attention_head_dim
        self.attention_cal = ProductAttentionLayer(use_scale=use_scale, mask_mod=atten_mask_mod)
        self.seed = seed
        self.use_res = use_res
        self.use_ln = use_ln
        self.head_concat = head_concat
        self.supports_masking = supports_masking
        self.ln_epsilon = 1e-3  # tf.keras.layers.LayerNormalization() default

    def build(self, input_shape):
        shape = [input_shape[-1], self.attention_head_dim, self.attention_dim]
        self.query_w = self.add_weight("query_w", shape, "glorot_uniform", seed=self.seed)
        self.key_w = self.add_weight("key_w", shape, "glorot_uniform", seed=self.seed)
        self.value_w = self.add_weight("value_w", shape, "glorot_uniform", seed=self.seed)  # unused, as in the reference
        if self.use_res:
            self.res_w = self.add_weight("res_w", shape, "glorot_uniform")
        self.ln_gamma = self.add_weight("ln_gamma", [self.attention_dim], "ones")
        self.ln_beta = self.add_weight("ln_beta", [self.attention_dim], "zeros")
        super().build(input_shape)

    def _args(self):
        return (self.query_w, self.key_w, self.res_w if self.use_res else None,
                self.ln_gamma if self.use_ln else None, self.ln_beta if self.use_ln else None)

    def fits_fused_kernel(self, inputs):
        """The fused kernel's menu (include/fil.h): K <= 64, A <= 16, H <= 8, F <= 512.  A reference-legal layer outside it
        (e.g. attention_dim = 32) takes the composed path below instead of raising."""
        return (inputs.dim() == 3 and inputs.shape[2] <= 64 and self.attention_dim <= 16 and self.attention_head_dim <= 8
                and inputs.shape[1] <= 512)

    def _composed(self, inputs, mask):
        """The reference's op order (:358-369) with the attention core on the stand-alone kernel (fil_pattn_*: A <= 64, any mask) --
        projections as tensordot, ProductAttentionLayer([q, k, v], mask), LayerNormalization (epsilon 1e-3) by torch."""
        Wq, Wk, Wr, g, b = self._args()
        q = torch.tensordot(inputs, Wq, dims=1).permute(2, 0, 1, 3)
        k = torch.tensordot(inputs, Wk, dims=1).permute(2, 0, 1, 3)
        atten_v = self.attention_cal([q, k, k], mask=mask)                  # v is projected with key_w (:360)
        res = torch.tensordot(inputs, Wr, dims=1).permute(2, 0, 1, 3) if Wr is not None else None
        if g is not None:
            atten_v = torch.nn.functional.layer_norm(atten_v, (self.attention_dim,), g, b, self.ln_epsilon)
        return atten_v, res

    def fused_relu(self, inputs):
        """relu(res + LN(attention)) in one kernel: what DnnLayer(res_unit=1, other_dense=[self]) computes."""
        Wq, Wk, Wr, g, b = self._args()
        if not self.fits_fused_kernel(inputs):
            atten_v, res = self._composed(inputs, None)
            return torch.relu(atten_v + res) if res is not None else torch.relu(atten_v)
        return Fn.autoint_interact(inputs, Wq, Wk, Wr, g, b, use_scale=self.attention_cal.use_scale, eps=self.ln_epsilon,
                                   precision=self.precision)

    def call(self, inputs, mask=None, **kwargs):
        Wq, Wk, Wr, g, b = self._args()
        if mask is None and self.fits_fused_kernel(inputs):
            atten_v, res = Fn.mult_head_attention(inputs, Wq, Wk, Wr, g, b, use_scale=self.attention_cal.use_scale,
                                                  eps=self.ln_epsilon, precision=self.precision)
        else:
            # masked, or a shape outside the fused kernel's menu: the composed path
            atten_v, res = self._composed(inputs, mask)
        if res is None:
            res = []
        if self.head_concat:
            atten_v = atten_v.permute(1, 0, 2, 3)
        if self.attention_head_dim == 1:
            return atten_v.squeeze(0)
        return [atten_v, res]
