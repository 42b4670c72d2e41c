cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_models_gpu.py tests/test_data_prepare.py -x -q -m gpu -k "embed or model or xdeepfm or deepfm or train or adam or data_prepare" 2>&1 | tail -3
for w in xdeepfm deepfm; do
timeout 300 python bench.py --workload $w --graph --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w eager %.4f graph %.4f'%(d['ms_per_step'], d['hipgraph_replay_ms_per_step']))"
done
