cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
bash tools/gpu_r5b.sh
VARIANTS="${STV}" bash tools/gpu_r5c.sh 2>&1 | grep -v amdgpu.ids
