#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_models_gpu.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "embed or model or zoo or xdeepfm or deepfm or train or sparse" 2>&1 | tail -4
python3 bench.py --workload xdeepfm --graph --steps 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('xdeepfm', j['ms_per_step'], j['hipgraph_replay_ms_per_step'])"
python3 bench.py --workload deepfm --graph --steps 20 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('deepfm', j['ms_per_step'], j['hipgraph_replay_ms_per_step'])"
