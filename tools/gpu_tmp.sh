cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_properties.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_attn.sh 2>&1 | tail -7
