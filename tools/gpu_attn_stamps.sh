cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/attn_stamps
for v in ${VARIANTS}; do
  export FIL_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libfil_$v.so
  for K in 64 16; do
    echo "== $v K=$K"
    timeout 200 python tools/attn_stamps.py 1 f16_mfma $K 2>&1 | tail -10
  done
done
