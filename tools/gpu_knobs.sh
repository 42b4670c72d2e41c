# the CIN parity subset under every launch-merging knob turned off (the fallbacks behind FIL_CIN_*=0 stay green): gpurun -- 'bash tools/gpu_knobs.sh'
# (round 4: 214 passed under each of HEADFOLD / FWDQ / DZ2 / QMERGE = 0; under QTAIL = 0 every parity check passes and the one test that
#  asserts the quadratic tail runs OTHER kernels than the fused tail -- test_cin_fused_tail_is_the_default_where_it_pays -- fails, as it must)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/knobs
for kn in FIL_CIN_HEADFOLD FIL_CIN_FWDQ FIL_CIN_DZ2 FIL_CIN_QMERGE FIL_CIN_QTAIL; do
  env $kn=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_cin and not benchmark_shape and not large_batch" > gpurun_out/knobs/$kn.log 2>&1
  echo "$kn=0: $(grep -v '^$' gpurun_out/knobs/$kn.log | tail -1)"
done
