"""CPU oracle for the feature-interaction hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the four reference layers
on the hot path:

  A1  InnerLayer / FmLayer          kon/model/ctr_model/layer/interactive_layer/interactive_layer.py:59-66,161-170
  A2  CrossLayer (DCN)              .../interactive_layer.py:264-282
  A3  CIN (xDeepFM)                 .../interactive_layer.py:306-327
  A4  MultHeadAttentionLayer +      kon/model/ctr_model/layer/behavior_layer/behavior_layer.py:292-311,335-377
      ProductAttentionLayer, wrapped by DnnLayer   kon/model/ctr_model/layer/core_layer/core_layer.py:201-226
  A5  packaging glue (StackLayer / Concatenate)    .../core_layer.py:49-55
  N1  SparseEmbed gather + LabelEncoder field-index work  .../interactive_layer.py:225-242, kon/utils/data_prepare.py:91-93

Two independent restatements are kept so that they can be checked against each
other:

  * ``oracle.graph``   -- op-for-op restatement of the reference TF2 graph in
    torch-CPU (same op decomposition TF would execute, including the
    materialised CIN outer product Z and its two transposes); dtype-parametric
    (fp64 = truth for tolerances and gradients via autograd, fp32 = the CPU
    baseline timed by bench.py's ``cpu_baseline`` leg).
  * ``oracle.closed``  -- closed-form NumPy fp64 formulas (einsum) of SURVEY.md
    Appendix A, with hand-derived backward formulas.

PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures for
this path, its arithmetic lives in TensorFlow 2.1, which is not
vendored under /root/reference and is not installable in this image, and the
reference's own modules cannot be imported (they import lightgbm / gensim /
seaborn at module load).  The oracle is therefore pinned only by (i) the two
restatements agreeing with each other, (ii) hand-derivable known-answer tests
(tests/test_oracle_kat.py) and (iii) for the field-index work, goldens produced
by the real third-party code the reference calls (sklearn LabelEncoder), see
tests/golden/make_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under ml_function_amd/ imports it; the product path has
no CPU fallback and raises if the HIP library is missing.
"""
