cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dcn" 2>&1 | grep -v "^$" | tail -25
