cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/full/pytest_gpu.log 2>&1
tail -15 gpurun_out/full/pytest_gpu.log
