"""ml_function_amd -- MI355X-native feature-interaction layers (FM, DCN cross, xDeepFM CIN, AutoInt interacting layer).

The Keras-style layer classes in ``ml_function_amd.layers`` keep the reference's constructor kwargs / build() / call()
surface; per-batch arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of include/fil.h
(libfil_hip.so, built in-tree by ``python -m ml_function_amd.build``).  There is no CPU fallback.
"""
__version__ = "0.1.0"
