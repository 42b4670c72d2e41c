"""Losses the reference compiles its models with (`model.compile(loss="binary_crossentropy")`), as HIP launches: the handful of
[B]-sized elementwise kernels torch spends on clamp + BCE + mean and their backward become one launch forward (which also leaves
d loss / d p) and one multiply backward.  GPU only, like every product path here."""
from .functional import binary_crossentropy

__all__ = ["binary_crossentropy"]
