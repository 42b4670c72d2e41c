cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_models_gpu.py tests/test_golden.py -x -q -m gpu -k "embed or zoo or adam or sparse or xdeepfm or deepfm" 2>&1 | tail -15
timeout 300 python examples/train_ctr.py --model XDeepFM --steps 40 --batch 1024 2>&1 | tail -4
timeout 300 python bench.py --workload deepfm --graph --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-400
