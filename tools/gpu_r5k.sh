cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "230 or 300-64 or test_attn_two_waves" > gpurun_out/r5k/test.log 2>&1
tail -6 gpurun_out/r5k/test.log
